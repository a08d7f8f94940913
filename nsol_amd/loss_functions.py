"""Robust loss functions rho(f^2) (drop-in for nsol/loss_functions.py:17-266).

Inside the solvers the loss is evaluated by the fused HIP kernel
nsol_loss_cost_grad_* (ops.loss_cost_grad).  The static methods below keep the
reference's element-wise API; they evaluate on the GPU as well (host arrays
make a round trip) through nsol_loss_eval_*.
"""
import numpy as np

from . import ops
from .device import is_device_tensor, to_device, to_numpy

_NAMES = ("linear", "soft_l1", "huber", "cauchy", "arctan")


def _eval(name, f2, f_scale, which, gamma=1.345):
    if is_device_tensor(f2):
        rho, drho = ops.loss_eval(f2.contiguous().view(-1), name, f_scale,
                                  gamma)
        return (rho if which == 0 else drho).view(f2.shape)
    arr = np.asarray(f2, dtype=np.float64)
    rho, drho = ops.loss_eval(to_device(arr.reshape(-1), np.float64), name,
                              f_scale, gamma)
    return to_numpy(rho if which == 0 else drho).reshape(arr.shape)


def _make(name, which):
    if name == "huber":       # loss_functions.py:148,169: gamma comes second
        def fn(f2, gamma=1.345, f_scale=1.):
            return _eval(name, f2, f_scale, which, gamma)
    else:
        def fn(f2, f_scale=1.):
            return _eval(name, f2, f_scale, which)
    fn.__name__ = ("gradient_" if which else "") + name
    return fn


class LossFunctions(object):

    get_loss = {n: _make(n, 0) for n in _NAMES}
    get_gradient_loss = {n: _make(n, 1) for n in _NAMES}

    @staticmethod
    def get_ell2_cost_from_residual(f, loss="linear", f_scale=1.):
        # loss_functions.py:31-35
        r = f if is_device_tensor(f) else to_device(
            np.asarray(f, dtype=np.float64).reshape(-1), np.float64)
        return ops.loss_cost_grad(r.contiguous().view(-1), loss, f_scale,
                                  want_grad=False)[0]

    @staticmethod
    def get_gradient_ell2_cost_from_residual(f, jac_f, loss="linear",
                                             f_scale=1.):
        # loss_functions.py:53-59: J^T (rho'(f^2) f) with a dense host Jacobian
        f = np.asarray(f, dtype=np.float64)
        _, g = ops.loss_cost_grad(to_device(f.reshape(-1), np.float64), loss,
                                  f_scale)
        return np.sum(to_numpy(g)[:, np.newaxis] * np.asarray(jac_f), axis=0)


for _n in _NAMES:
    setattr(LossFunctions, _n, staticmethod(LossFunctions.get_loss[_n]))
    setattr(LossFunctions, "gradient_" + _n,
            staticmethod(LossFunctions.get_gradient_loss[_n]))
