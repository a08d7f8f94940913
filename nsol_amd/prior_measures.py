"""Regulariser values for reporting (drop-in for nsol/prior_measures.py:17-52),
evaluated with HIP reductions.  x: NumPy array or device tensor (flat);
D: a gradient callable as handed to the solvers."""
import numpy as np

from . import ops
from .bridge import BridgedCallable
from .device import is_device_tensor, to_device


def _dev(x):
    if is_device_tensor(x):
        return x.contiguous().view(-1)
    return to_device(np.asarray(x, dtype=np.float64).reshape(-1), np.float64)


def _apply(D, x):
    x = _dev(x)
    return BridgedCallable(D, np.float32 if "32" in str(x.dtype)
                           else np.float64)(x)


class PriorMeasures(object):

    @staticmethod
    def zeroth_order_tikhonov(x):
        x = _dev(x)
        return 0.5 * ops.dot(x, x)

    @staticmethod
    def first_order_tikhonov(x, D):
        g = _apply(D, x)
        return 0.5 * ops.dot(g, g)

    @staticmethod
    def total_variation(x, D, dimension):
        return ops.vector_norm_sum(_apply(D, x), dimension, 0)

    @staticmethod
    def huber(x, D, dimension, gamma=0.05):
        return ops.vector_norm_sum(_apply(D, x), dimension, 1, gamma)
