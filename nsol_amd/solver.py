"""Solver base class (drop-in for nsol/solver.py:21-174).

State lives in HBM as torch HIP tensors of the working dtype (float32 by
default, float64 for validation; the reference is float64 throughout,
solver.py:37).  `x0` may be a NumPy array (as in the reference) or a 1-D torch
HIP tensor that is already resident.
"""
import datetime
import time

import numpy as np

from . import ops
from ._accessors import add_accessors
from .device import (get_default_dtype, is_device_tensor, to_device, to_numpy,
                     torch_dtype)


class Solver(object):

    def __init__(self, x0, x_scale, verbose, dtype=None, _borrow=False):
        self._dtype = np.dtype(dtype or get_default_dtype()).type
        if self._dtype not in (np.float32, np.float64):
            raise ValueError("dtype must be float32 or float64")
        self._x_scale = float(x_scale)
        # _borrow: the caller (an outer solver of this package) owns the device
        # tensors it passes and will not touch them while this solver runs; with
        # x_scale = 1 they are then used as they are instead of being copied
        # through a division by one
        self._borrow = bool(_borrow)
        self._verbose = verbose
        self._computational_time = datetime.timedelta(seconds=0)
        self._observer = None
        self._set_x0(x0)

    # x0 is kept as given (a host copy in its own float dtype, or device) and
    # uploaded lazily so that solvers can be constructed and inspected on a
    # machine without a GPU.  The division by x_scale (solver.py:37) runs on the
    # device in the working precision -- the same way the data term is scaled,
    # so x0 = b gives bit-equal scaled arrays -- instead of two float64 passes
    # over the host copy (0.14 s at 512^3).
    def _borrowed(self, v):
        """v itself (flat) when it may be used without a private copy."""
        if self._borrow and self._x_scale == 1.0 and is_device_tensor(v) and \
                v.dtype == torch_dtype(self._dtype) and v.is_contiguous():
            return v.view(-1)
        return None

    def _set_x0(self, x0):
        if is_device_tensor(x0):
            self._x0_ndim = x0.dim()
            self._x0_host = None
            self._x0_dev = self._borrowed(x0)
            if self._x0_dev is None:
                # (never written in place by the solvers; a start vector that a
                # caller hands to one solver after the other is divided once)
                from .proximal_operators import scaled_tensor
                self._x0_dev = scaled_tensor(x0, self._x_scale,
                                             torch_dtype(self._dtype))
        else:
            arr = np.asarray(x0)
            keep = arr.dtype if arr.dtype in (np.float32, np.float64) \
                else np.float64
            self._x0_ndim = arr.ndim
            self._x0_host = np.array(arr, dtype=keep)     # private, unscaled copy
            self._x0_dev = None
        self._x = None

    def _x0_device(self):
        if self._x0_dev is None:
            self._x0_dev = ops.scale(
                to_device(self._x0_host.reshape(-1), self._dtype),
                self._x_scale, divide=True)
        return self._x0_dev

    def set_x0(self, x0):
        self._set_x0(x0)

    def get_x0(self):
        if self._x0_host is not None:
            return np.array(self._x0_host, dtype=np.float64)
        return to_numpy(ops.scale(self._x0_dev, self._x_scale))

    def get_x_device(self):
        """Current iterate times x_scale as a NEW flat device tensor."""
        cur = self._x if self._x is not None else self._x0_device()
        return ops.scale(cur, self._x_scale)

    def get_x(self):
        # solver.py:117-118: copy, multiplied by x_scale, float64 on the host
        if self._x is None and self._x0_host is not None:
            return np.array(self._x0_host, dtype=np.float64)
        return to_numpy(self.get_x_device())

    def run(self):
        if self._x0_ndim != 1:
            raise ValueError("Initial value x0 must be a 1D array")
        import torch
        t0 = time.time()
        self._run()
        if self._sync_after_run and not self._borrow:
            torch.cuda.synchronize()
            # (a persistent-kernel run that timed out is repeated here, before anyone
            # can consume its result on the device or on the host)
            ops.settle_persist_runs(synchronize=False)
        # (_borrow / _sync_after_run = False: an inner solve of an outer solver of this
        # package, which goes on enqueueing behind it and synchronises at the end of ITS
        # run: draining the device here left it idle while the outer loop prepared its
        # next step)
        self._computational_time = datetime.timedelta(
            seconds=time.time() - t0)
        if self._verbose:
            print("Required computational time: %s" %
                  (self.get_computational_time()))
        if self._observer is not None:
            self._observer.set_computational_time(
                self.get_computational_time())

    _sync_after_run = True

    def _run(self):
        raise NotImplementedError

    def print_statistics(self):
        raise NotImplementedError


add_accessors(Solver, ["x_scale", "verbose", "observer"])
add_accessors(Solver, ["computational_time", "dtype"], setters=False)
