"""Linear operators of the NSoL hot path on MI355X (drop-in for
nsol/linear_operators.py).

The factories keep the reference's names and return CALLABLE OBJECTS instead of
lambdas.  A callable accepts
  * an N-D NumPy array  -> uploads, runs the HIP kernel, returns a NumPy array
    of the same float dtype (float64 unless the input is float32);
  * an N-D torch HIP tensor (float32/float64) -> returns a device tensor, no
    host traffic (what the solvers use);
  * the solvers' symbolic probe (symbolic.Sym) -> describes itself so that
    PrimalDualSolver / ADMMLinearSolver can dispatch to fused kernels.
Shapes follow the reference: grad maps (Z,Y,X) -> (3Z,Y,X) (linear_operators.py:140).
"""
import numpy as np

from . import kernels as Kernels
from . import ops
from .device import is_device_tensor, to_device, to_numpy
from .symbolic import Sym, TraceAbort


def _stacked_shape(shape, dim):
    shape = tuple(shape)
    if dim == 1:
        return shape
    return (dim * shape[0],) + shape[1:]


def _unstacked_shape(shape, dim):
    shape = tuple(shape)
    if dim == 1:
        return shape
    if shape[0] % dim:
        raise ValueError("leading axis %d is not a multiple of %d" %
                         (shape[0], dim))
    return (shape[0] // dim,) + shape[1:]


class DeviceOperator(object):
    """Common host/device/symbolic dispatch."""

    kind = "operator"

    def _out_shape(self, in_shape):
        return tuple(in_shape)

    def _apply(self, x, in_shape):  # x: flat contiguous device tensor
        raise NotImplementedError

    def _describe(self, in_shape):
        return (self.kind, self, tuple(in_shape))

    def __call__(self, x):
        if isinstance(x, Sym):
            if x.desc is not None:
                raise TraceAbort("composition of operators is not fused")
            return Sym(self._out_shape(x.shape), self._describe(x.shape))
        if is_device_tensor(x):
            shape = tuple(x.shape)
            out = self._apply(x.contiguous().view(-1), shape)
            return out.view(self._out_shape(shape))
        arr = np.asarray(x)
        dt = arr.dtype.type if arr.dtype in (np.float32, np.float64) \
            else np.float64
        out = self._apply(to_device(arr, dt).view(-1), arr.shape)
        return to_numpy(out, dt).reshape(self._out_shape(arr.shape))


class _DimChecked(DeviceOperator):

    def __init__(self, dimension, spacing):
        self.dimension = dimension
        self.spacing = np.atleast_1d(spacing).astype(float)
        self.w = ops.inv_spacing(self.spacing, dimension)

    def _check(self, shape, stacked=False):
        if len(shape) != self.dimension:
            raise RuntimeError(
                "%dD operator applied to an array with %d axes" %
                (self.dimension, len(shape)))


class GradientOperator(_DimChecked):
    """K = nabla, zero-padded forward differences stacked on axis 0
    (linear_operators.py:121-144, mode="constant")."""
    kind = "grad"

    def _out_shape(self, in_shape):
        return _stacked_shape(in_shape, self.dimension)

    def _apply(self, x, in_shape):
        self._check(in_shape)
        return ops.grad(x, in_shape, self.w)


class GradientAdjointOperator(_DimChecked):
    """K^T (linear_operators.py:158-169)."""
    kind = "grad_adj"

    def _out_shape(self, in_shape):
        return _unstacked_shape(in_shape, self.dimension)

    def _apply(self, p, in_shape):
        self._check(in_shape)
        return ops.grad_adj(p, _unstacked_shape(in_shape, self.dimension),
                            self.w)


class AxisDifferenceOperator(_DimChecked):
    """D_a or D_a^T along one direction (linear_operators.py:98-106)."""
    kind = "diff"

    def __init__(self, dimension, spacing, direction, adjoint):
        _DimChecked.__init__(self, dimension, spacing)
        self.direction = direction
        self.adjoint = adjoint

    def _describe(self, in_shape):
        return (self.kind, self, tuple(in_shape))

    def _apply(self, x, in_shape):
        self._check(in_shape)
        return ops.diff_axis(x, in_shape, self.direction, self.adjoint,
                             self.w[self.direction])


def _ndimage_convolve_params(kernel):
    """scipy.ndimage.convolve(x, k) == correlate(x, flip(k)) with the centre
    of the flipped kernel at size//2 (odd) or size//2-1 (even)."""
    flipped = kernel[tuple([slice(None, None, -1)] * kernel.ndim)]
    centre = [s // 2 - (1 if s % 2 == 0 else 0) for s in kernel.shape]
    return np.ascontiguousarray(flipped, dtype=np.float64), centre


def _rank1_factors(kernel, tol=1e-13):
    """Per-axis factors if the tap array is an outer product, else None."""
    if kernel.ndim == 1:
        return [kernel]
    total = kernel.sum()
    if total == 0:
        return None
    facs = [kernel.sum(axis=tuple(a for a in range(kernel.ndim) if a != ax))
            for ax in range(kernel.ndim)]
    rebuilt = facs[0]
    for f in facs[1:]:
        rebuilt = np.multiply.outer(rebuilt, f)
    rebuilt = rebuilt / total ** (kernel.ndim - 1)
    if np.max(np.abs(rebuilt - kernel)) > tol * np.max(np.abs(kernel)):
        return None
    facs[0] = facs[0] / total ** (kernel.ndim - 1)
    return facs


# one-pass 3-D blur (nsol_corr3_wrap_*); False = always three 1-D passes
USE_FUSED_BLUR3 = True
# LSMR's top-block update as the epilogue of that blur (nsol_corr3_wrap_axpby_*)
USE_BLUR_EPILOGUE = True


class ConvolutionOperator(DeviceOperator):
    """x -> scipy.ndimage.convolve(x, kernel, mode) on the GPU
    (linear_operators.py:60-68).  Separable kernels (every Gaussian with a
    diagonal covariance) run as 1-D passes, anything else as dense taps."""
    kind = "conv"

    def __init__(self, dimension, kernel, mode="wrap"):
        kernel = np.asarray(kernel, dtype=np.float64)
        if kernel.ndim != dimension:
            raise RuntimeError("filter weights array has incorrect shape.")
        if mode not in ops.MODES:
            raise RuntimeError("boundary mode not supported")
        self.dimension = dimension
        self.kernel = kernel
        self.mode = mode
        flipped, centre = _ndimage_convolve_params(kernel)
        self._flipped, self._centre = flipped, centre
        facs = _rank1_factors(flipped)
        if facs is not None and max(f.size for f in facs) <= 129:
            self._passes = [(ax + 3 - dimension, facs[ax], centre[ax])
                            for ax in range(dimension)
                            if not (facs[ax].size == 1 and facs[ax][0] == 1.0)]
        else:
            self._passes = None
        self._taps_dev = {}

    @property
    def separable(self):
        return self._passes is not None

    def _fusable3(self):
        """Three periodic passes with the same odd tap count, centred: one
        launch of nsol_corr3_wrap_* instead of three of nsol_corr_axis_*."""
        p = self._passes
        return (self.dimension == 3 and self.mode == "wrap" and len(p) == 3
                and [a for a, _, _ in p] == [0, 1, 2]
                and len({t.size for _, t, _ in p}) == 1
                and p[0][1].size % 2 == 1
                and all(c == t.size // 2 for _, t, c in p))

    def apply_axpby(self, x, io, in_shape, ca, cb, result=None):
        """io = ca * A(x) + cb * io in place with the sum of squares of the
        result (flat device tensors; the top block of LSMR's u update as the
        epilogue of the one-pass blur).  None when that kernel does not apply.
        result: the caller's one-element float64 device tensor for the sum (then
        nothing is read back)."""
        if not (USE_FUSED_BLUR3 and USE_BLUR_EPILOGUE and self._passes and
                len(in_shape) == 3 and self._fusable3()):
            return None
        return ops.corr3_wrap_axpby(x, io, in_shape, self._passes[0][1],
                                    self._passes[1][1], self._passes[2][1], ca, cb,
                                    result=result)

    def apply_norms(self, x, out, in_shape, w, result):
        """out = A(x) with result[0] = sum out^2 and result[1] = sum |grad x|^2 of
        the input (w: the gradient's weights), both from the one-pass blur; None
        when that kernel does not apply."""
        if not (USE_FUSED_BLUR3 and USE_BLUR_EPILOGUE and self._passes and
                len(in_shape) == 3 and self._fusable3()):
            return None
        return ops.corr3_wrap_norms(x, out, in_shape, self._passes[0][1],
                                    self._passes[1][1], self._passes[2][1], w, result)

    def lanczos_halves(self, in_shape):
        """(half_a, half_b) -- ops.corr3_lanczos_a / _b bound to this blur's taps -- when
        the one-pass blur can take both halves of a Lanczos step on A'A + rho B'B itself
        (symmetric separable taps on a 3-D grid), else None."""
        if not (USE_FUSED_BLUR3 and USE_BLUR_EPILOGUE and self._passes and
                len(in_shape) == 3 and self._fusable3()):
            return None
        tz, ty, tx = (self._passes[0][1], self._passes[1][1], self._passes[2][1])

        lean = bool(ops.LEAN_LANCZOS_HALVES)
        held = {}

        def half_a(y, y_prev, t, q0, lb, step):
            if lean:
                # (q0 is not formed: the second half takes y_prev itself)
                held["y_prev"] = y_prev
                return ops.corr3_lanczos_a2(y, t, in_shape, tz, ty, tx, lb, step)
            return ops.corr3_lanczos_a(y, y_prev, t, q0, in_shape, tz, ty, tx, lb, step)

        def half_b(t, q0, y, y_new, lb, step):
            if lean:
                return ops.corr3_lanczos_b2(t, y, held.get("y_prev"), y_new, in_shape, tz,
                                            ty, tx, lb, step)
            return ops.corr3_lanczos_b(t, q0, y, y_new, in_shape, tz, ty, tx, lb, step)
        half_a.lean = lean
        return half_a, half_b

    def apply_loss(self, x, b, in_shape, loss, f_scale, result):
        """rho'(r^2) r for r = A x - b with 1/2 sum rho(r^2) in the device slot `result`,
        from the one-pass blur (ops.corr3_wrap_loss); None where it does not apply."""
        if not (USE_FUSED_BLUR3 and USE_BLUR_EPILOGUE and self._passes and
                len(in_shape) == 3 and self._fusable3()):
            return None
        return ops.corr3_wrap_loss(x, b, in_shape, self._passes[0][1], self._passes[1][1],
                                   self._passes[2][1], loss, f_scale, result)

    def _apply(self, x, in_shape):
        if len(in_shape) != self.dimension:
            raise RuntimeError("%dD convolution applied to %d axes" %
                               (self.dimension, len(in_shape)))
        if self._passes is not None:
            if not self._passes:
                return x.clone()
            if USE_FUSED_BLUR3 and self._fusable3():
                res = ops.corr3_wrap(x, in_shape, self._passes[0][1],
                                     self._passes[1][1], self._passes[2][1])
                if res is not None:
                    return res
            cur = x
            for axis3, taps, centre in self._passes:
                cur = ops.corr_axis(cur, in_shape, axis3, taps, centre,
                                    self.mode)
            return cur
        key = (x.dtype, x.device.index)
        if key not in self._taps_dev:
            self._taps_dev[key] = to_device(
                self._flipped.reshape(-1),
                np.float32 if "32" in str(x.dtype) else np.float64)
        k3 = (1,) * (3 - self.dimension) + self._flipped.shape
        c3 = (0,) * (3 - self.dimension) + tuple(self._centre)
        return ops.corr_dense(x, in_shape, self._taps_dev[key], k3, c3,
                              self.mode)


class LinearOperators(object):

    def __init__(self, dimension, spacing):
        self._dimension = dimension
        self._spacing = spacing
        self._kernels = {1: Kernels.Kernels1D, 2: Kernels.Kernels2D,
                         3: Kernels.Kernels3D}[dimension](spacing=spacing)

    def get_spacing(self):
        return self._spacing

    def get_dimension(self):
        return self._dimension

    def get_convolution_and_adjoint_convolution_operators(
            self, kernel, mode="wrap"):
        # the reference uses the SAME kernel for the adjoint
        # (linear_operators.py:63); exact for symmetric taps
        A = ConvolutionOperator(self._dimension, kernel, mode)
        return A, A

    def get_gaussian_blurring_operators(self, cov, alpha_cut=3):
        kernel = self._kernels.get_gaussian(cov=cov, alpha_cut=alpha_cut)
        return self.get_convolution_and_adjoint_convolution_operators(kernel)

    def _axis_operators(self, direction, mode):
        if direction >= self._dimension:
            raise AttributeError("no such direction in %dD" % self._dimension)
        if mode == "constant":
            sp = self._kernels.get_spacing()
            return (AxisDifferenceOperator(self._dimension, sp, direction, 0),
                    AxisDifferenceOperator(self._dimension, sp, direction, 1))
        fwd = self._kernels._difference(direction, False)
        bwd = -self._kernels._difference(direction, True)
        return (ConvolutionOperator(self._dimension, fwd, mode),
                ConvolutionOperator(self._dimension, bwd, mode))

    def get_dx_operators(self, mode="constant"):
        return self._axis_operators(0, mode)

    def get_gradient_operators(self, mode="constant"):
        sp = self._kernels.get_spacing()
        if mode == "constant":
            return (GradientOperator(self._dimension, sp),
                    GradientAdjointOperator(self._dimension, sp))
        pairs = [self._axis_operators(a, mode)
                 for a in range(self._dimension)]
        return (_StackedOperator([p[0] for p in pairs]),
                _StackedAdjointOperator([p[1] for p in pairs]))


class _StackedOperator(DeviceOperator):
    """grad for non-default boundary modes: concatenation of per-axis passes."""
    kind = "stack"

    def __init__(self, parts):
        self.parts = parts

    def _out_shape(self, in_shape):
        return _stacked_shape(in_shape, len(self.parts))

    def _apply(self, x, in_shape):
        import torch
        return torch.cat([p._apply(x, in_shape) for p in self.parts])


class _StackedAdjointOperator(DeviceOperator):
    kind = "stack_adj"

    def __init__(self, parts):
        self.parts = parts

    def _out_shape(self, in_shape):
        return _unstacked_shape(in_shape, len(self.parts))

    def _apply(self, p, in_shape):
        d = len(self.parts)
        shp = _unstacked_shape(in_shape, d)
        m = p.numel() // d
        acc = self.parts[0]._apply(p[:m], shp)
        for a in range(1, d):
            acc = ops.lincomb2(1.0, acc, 1.0,
                               self.parts[a]._apply(p[a * m:(a + 1) * m], shp),
                               out=acc)
        return acc


class LinearOperators1D(LinearOperators):

    def __init__(self, spacing=1):
        LinearOperators.__init__(self, dimension=1, spacing=spacing)


class LinearOperators2D(LinearOperators):

    def __init__(self, spacing=np.ones(2)):
        LinearOperators.__init__(self, dimension=2, spacing=spacing)

    def get_dy_operators(self, mode="constant"):
        return self._axis_operators(1, mode)


class LinearOperators3D(LinearOperators2D):

    def __init__(self, spacing=np.ones(3)):
        LinearOperators.__init__(self, dimension=3, spacing=spacing)

    def get_dz_operators(self, mode="constant"):
        return self._axis_operators(2, mode)
