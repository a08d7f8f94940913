"""Base class of the linear least-squares solvers (drop-in for
nsol/linear_solver.py:25-344).  Data term  1/2 sum rho((A x - b)^2)."""
import numpy as np

from . import ops
from .bridge import BridgedCallable
from .device import is_device_tensor, to_device, to_numpy, torch_dtype
from .loss_functions import LossFunctions as lf
from .solver import Solver
from ._accessors import add_accessors


class _LazyScaled(object):
    """A host array standing for raw / x_scale; the quotient is formed on the
    device, in the working precision, the first time it is needed."""

    def __init__(self, raw, x_scale):
        arr = np.asarray(raw)
        keep = arr.dtype if arr.dtype in (np.float32, np.float64) \
            else np.float64
        self.raw = np.array(arr, dtype=keep).reshape(-1)   # private copy
        self.x_scale = float(x_scale)
        self._cache = {}

    def device(self, dtype):
        key = np.dtype(dtype).name
        if key not in self._cache:
            self._cache[key] = ops.scale(to_device(self.raw, dtype),
                                         self.x_scale, divide=True)
        return self._cache[key]


class LinearSolver(Solver):

    def __init__(self, A, A_adj, b, x0, alpha, x_scale, data_loss,
                 data_loss_scale, minimizer, iter_max, verbose, dtype=None,
                 _borrow=False):
        Solver.__init__(self, x0=x0, x_scale=x_scale, verbose=verbose,
                        dtype=dtype, _borrow=_borrow)
        self._A = A
        self._A_adj = A_adj
        self._b = self._scaled_data(b)            # linear_solver.py:73
        self._alpha = float(alpha)
        self._data_loss = data_loss
        self._data_loss_scale = float(data_loss_scale)
        self._minimizer = minimizer
        self._iter_max = iter_max

    def _scaled(self, v):
        """v / x_scale, kept on the side (host/device) it was given on."""
        if is_device_tensor(v):
            own = self._borrowed(v)
            if own is not None:
                return own
            return ops.scale(v.to(torch_dtype(self._dtype)).contiguous()
                             .view(-1), self._x_scale, divide=True)
        return np.asarray(v, dtype=np.float64) / self._x_scale

    def _scaled_data(self, v):
        """The observation b: a large host array is kept as given and divided by
        x_scale on the device at its first use (once; the device copy is cached)
        instead of in two float64 passes on the host at construction and one
        float64 upload per outer iteration."""
        if not is_device_tensor(v) and np.size(v) >= (1 << 20):
            return _LazyScaled(v, self._x_scale)
        if is_device_tensor(v) and self._borrowed(v) is None:
            # (a solver built once per outer iteration around the same b:
            # divided once, proximal_operators.scaled_tensor)
            from .proximal_operators import scaled_tensor
            return scaled_tensor(v, self._x_scale, torch_dtype(self._dtype))
        return self._scaled(v)

    def _dev(self, v):
        """Flat device tensor of the working dtype (uploads host arrays)."""
        if isinstance(v, _LazyScaled):
            return v.device(self._dtype)
        if is_device_tensor(v):
            return v.to(torch_dtype(self._dtype)).contiguous().view(-1)
        return to_device(np.asarray(v, dtype=np.float64).reshape(-1),
                         self._dtype)

    def get_b(self):
        if isinstance(self._b, _LazyScaled):
            return np.array(self._b.raw, dtype=np.float64)
        if is_device_tensor(self._b):
            return to_numpy(ops.scale(self._b, self._x_scale))
        return np.array(self._b) * self._x_scale

    def set_data_loss(self, data_loss):
        if data_loss not in lf.get_loss.keys():
            raise ValueError("data_loss must be in " +
                             str(lf.get_loss.keys()))
        self._data_loss = data_loss

    # ---- costs at the current iterate (linear_solver.py:242-312)
    def _current(self):
        return self._x if self._x is not None else self._x0_device()

    def get_total_cost(self):
        return self.get_cost_data_term() + \
            self._alpha * self.get_cost_regularization_term()

    def get_cost_data_term(self):
        return self._get_cost_data_term(self._current())

    def get_ell2_cost_data_term(self):
        return self._get_ell2_cost_data_term(self._current())

    def get_cost_regularization_term(self):
        return self._get_cost_regularization_term(self._current())

    def print_statistics(self, fmt="%.3e"):
        cost_data = self.get_cost_data_term()
        cost_data_ell2 = self.get_ell2_cost_data_term()
        cost_reg = self.get_cost_regularization_term()
        print("Summary Optimization")
        print("Computational time: %s" % (self.get_computational_time()))
        print("Cost data term (f, loss=%s, scale=%g): " %
              (self._data_loss, self._data_loss_scale) + fmt % cost_data +
              " (ell2-cost: " + fmt % cost_data_ell2 + ")")
        print("Cost regularization term (g): " + fmt % cost_reg)
        print("Total cost (f + alpha g; alpha = %g" % self._alpha + "): " +
              fmt % (cost_data + self._alpha * cost_reg))

    # ---- device evaluation of the data term (linear_solver.py:315-340)
    def _residual(self, x):
        A = BridgedCallable(self._A, self._dtype)
        return ops.lincomb2(1.0, A(x), -1.0, self._dev(self._b))

    def _get_cost_data_term(self, x):
        r = self._residual(self._dev(x))
        return ops.loss_cost_grad(r, self._data_loss, self._data_loss_scale,
                                  want_grad=False)[0]

    def _get_ell2_cost_data_term(self, x):
        r = self._residual(self._dev(x))
        return 0.5 * ops.dot(r, r)

    def _get_gradient_cost_data_term(self, x):
        r = self._residual(self._dev(x))
        _, g = ops.loss_cost_grad(r, self._data_loss, self._data_loss_scale,
                                  out=r)
        return BridgedCallable(self._A_adj, self._dtype)(g)

    def _get_cost_regularization_term(self, x):
        raise NotImplementedError


add_accessors(LinearSolver, ["A", "A_adj", "data_loss"], setters=False)
add_accessors(LinearSolver, ["alpha", "data_loss_scale", "minimizer",
                             "iter_max"])
