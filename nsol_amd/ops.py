"""Device-tensor front end of the C ABI (include/nsol_hip.h).

Every function takes/returns contiguous torch HIP tensors of float32 or
float64 and launches hand-written kernels from libnsol_hip.so on the current
torch stream.  No arithmetic is done by torch itself.
"""
import numpy as np
import torch

from . import _lib, _timing
from .device import suffix, stream_ptr, empty_like

MODES = {"constant": 0, "wrap": 1, "nearest": 2, "reflect": 3, "mirror": 4}
LOSSES = {"linear": 0, "soft_l1": 1, "huber": 2, "cauchy": 3, "arctan": 4}
PD_REG_TV, PD_REG_HUBER, PD_DATA_L2, PD_DATA_L1 = 0, 1, 0, 2


def _fn(name, t):
    fn = getattr(_lib.load(), "nsol_%s_%s" % (name, suffix(t)))
    kt = _timing.active()              # (measurement only: bench.py / bench_admm.py)
    return fn if kt is None else kt.wrap(name, fn)


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(t):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()):
        raise TypeError("expected a contiguous HIP device tensor")
    return t


_bump = torch.autograd.graph.increment_version


def _wrote(first, *more):
    """The kernels write through raw pointers, which torch cannot see: count the
    write on the tensors' version counters (views share their base's).  That counter
    is what the caches of b / x_scale, A^T b and |b|^2 are keyed on
    (nsol_amd/_caches.py).  Returns its first argument."""
    if first is not None:
        _bump(first)
    for t in more:
        if t is not None:
            _bump(t)
    return first


def _same(x, *others):
    """Multi-operand kernels take their launch geometry and element type from
    the first operand: every other operand must be a contiguous device tensor
    of the same dtype and length, or the kernel would read out of bounds /
    reinterpret the bytes."""
    _chk(x)
    for t in others:
        _chk(t)
        if t.dtype != x.dtype or t.numel() != x.numel():
            raise ValueError(
                "operand mismatch: %s[%d] against %s[%d]" %
                (str(t.dtype), t.numel(), str(x.dtype), x.numel()))
    return x


def dims3(shape):
    """(ndim, nz, ny, nx) of an N-D volume shape (N = 1, 2, 3)."""
    shape = tuple(int(s) for s in shape)
    nd = len(shape)
    if nd == 1:
        return 1, 1, 1, shape[0]
    if nd == 2:
        return 2, 1, shape[0], shape[1]
    if nd == 3:
        return 3, shape[0], shape[1], shape[2]
    raise ValueError("only 1-D, 2-D and 3-D volumes are supported")


def inv_spacing(spacing, ndim):
    """(wx, wy, wz) = 1/spacing; spacing[0] belongs to the LAST array axis
    (reference kernels.py:240-286)."""
    sp = np.atleast_1d(spacing).astype(float)
    w = [1.0, 1.0, 1.0]
    for a in range(ndim):
        w[a] = 1.0 / sp[a]
    return tuple(w)


# ------------------------------------------------------------ operators ----
def grad(x, shape, w, out=None):
    _chk(x)
    ndim, nz, ny, nx = dims3(shape)
    if out is None:
        out = empty_like(x, ndim * x.numel())
    _lib.check(_fn("grad", x)(_p(x), _p(out), ndim, nz, ny, nx, w[0], w[1],
                              w[2], stream_ptr()), "nsol_grad")
    return _wrote(out)


def grad_adj(p, shape, w, out=None):
    _chk(p)
    ndim, nz, ny, nx = dims3(shape)
    if out is None:
        out = empty_like(p, nz * ny * nx)
    _lib.check(_fn("grad_adj", p)(_p(p), _p(out), ndim, nz, ny, nx, w[0], w[1],
                                  w[2], stream_ptr()), "nsol_grad_adj")
    return _wrote(out)


def grad_adj_axpy(p, x, tau, shape, w, out=None):
    """x - tau * grad_adj(p) in one pass."""
    _chk(p)
    _chk(x)
    ndim, nz, ny, nx = dims3(shape)
    if p.dtype != x.dtype or x.numel() != nz * ny * nx or \
            p.numel() != ndim * x.numel():
        raise ValueError("operand mismatch: p %s[%d], x %s[%d] for shape %r" %
                         (str(p.dtype), p.numel(), str(x.dtype), x.numel(),
                          tuple(shape)))
    if out is None:
        out = empty_like(x)
    _lib.check(_fn("grad_adj_axpy", p)(_p(p), _p(x), _p(out), ndim, nz, ny, nx,
                                       w[0], w[1], w[2], float(tau),
                                       stream_ptr()), "nsol_grad_adj_axpy")
    return _wrote(out)


def extrapolate(a, b, theta, out=None):
    """a + theta * (a - b) (the over-relaxation step, the reference's rounding)."""
    _same(a, b)
    if out is None:
        out = empty_like(a)
    _lib.check(_fn("extrapolate", a)(_p(out), _p(a), _p(b), float(theta),
                                     a.numel(), stream_ptr()), "nsol_extrapolate")
    return _wrote(out)


def diff_axis(x, shape, direction, adjoint, w):
    _chk(x)
    _, nz, ny, nx = dims3(shape)
    out = empty_like(x)
    _lib.check(_fn("diff_axis", x)(_p(x), _p(out), int(direction),
                                   int(bool(adjoint)), nz, ny, nx, float(w),
                                   stream_ptr()), "nsol_diff_axis")
    return _wrote(out)


def corr_axis(x, shape, axis3, taps, centre, mode, out=None):
    """1-D correlation along axis3 (0=z,1=y,2=x of the padded 3-D shape)."""
    _chk(x)
    _, nz, ny, nx = dims3(shape)
    taps = np.ascontiguousarray(taps, dtype=np.float64)
    if out is None:
        out = empty_like(x)
    _lib.check(_fn("corr_axis", x)(
        _p(x), _p(out), int(axis3), nz, ny, nx, taps.ctypes.data,
        int(taps.size), int(centre), MODES[mode], stream_ptr()),
        "nsol_corr_axis")
    return _wrote(out)


def corr3_wrap(x, shape, taps_z, taps_y, taps_x, out=None):
    """Separable periodic 3-D correlation in one pass (passes x, y, z); returns
    None when the fused kernel does not apply (nothing was launched)."""
    _chk(x)
    _, nz, ny, nx = dims3(shape)
    tz, ty, tx = (np.ascontiguousarray(t, dtype=np.float64)
                  for t in (taps_z, taps_y, taps_x))
    if not (tz.size == ty.size == tx.size):
        return None
    if out is None:
        out = empty_like(x)
    rc = _fn("corr3_wrap", x)(_p(x), _p(out), nz, ny, nx, tz.ctypes.data,
                              ty.ctypes.data, tx.ctypes.data, int(tz.size),
                              stream_ptr())
    if rc == -2:
        return None
    _lib.check(rc, "nsol_corr3_wrap")
    return _wrote(out)


def corr3_wrap_axpby(x, io, shape, taps_z, taps_y, taps_x, ca, cb, sync=True,
                     result=None):
    """io = ca * blur(x) + cb * io in place (the blur's epilogue; A x never goes
    to memory); returns the sum of squares of the new io, or None when the
    LDS-DMA staged kernel does not apply (nothing was launched).  result: a
    one-element float64 device tensor of the caller's that receives the sum
    instead (returned as it is, not read back)."""
    _same(x, io)
    _, nz, ny, nx = dims3(shape)
    tz, ty, tx = (np.ascontiguousarray(t, dtype=np.float64)
                  for t in (taps_z, taps_y, taps_x))
    if not (tz.size == ty.size == tx.size):
        return None
    ws, res = _workspace(x.device)
    if result is not None:
        res, sync = result, False
    rc = _fn("corr3_wrap_axpby", x)(
        _p(x), _p(io), nz, ny, nx, tz.ctypes.data, ty.ctypes.data, tx.ctypes.data,
        int(tz.size), float(ca), float(cb), _p(res), _p(ws), int(ws.numel()),
        stream_ptr())
    if rc == -2:
        return None
    _lib.check(rc, "nsol_corr3_wrap_axpby")
    _wrote(io)
    return float(res.item()) if sync else res


def corr3_wrap_norms(x, out, shape, taps_z, taps_y, taps_x, w, result):
    """out = blur(x) with result[0] = sum out^2 and result[1] = sum |grad x|^2 of the
    INPUT (weights w = inverse spacings as for tk1_grad_norm), both taken by the blur
    itself; result: a two-element float64 device tensor of the caller's (returned, not
    read back).  None when that kernel does not apply (nothing was launched)."""
    _same(x, out)
    _chk(result)
    if result.numel() != 2 or str(result.dtype) != "torch.float64":
        raise ValueError("corr3_wrap_norms: result must hold two float64 values")
    ndim, nz, ny, nx = dims3(shape)
    tz, ty, tx = (np.ascontiguousarray(t, dtype=np.float64)
                  for t in (taps_z, taps_y, taps_x))
    if nz * ny * nx != x.numel():
        raise ValueError("corr3_wrap_norms: %d elements for shape %r" %
                         (x.numel(), tuple(shape)))
    if ndim != 3 or not (tz.size == ty.size == tx.size):
        return None
    ws, _ = _workspace(x.device)
    rc = _fn("corr3_wrap_norms", x)(
        _p(x), _p(out), nz, ny, nx, tz.ctypes.data, ty.ctypes.data, tx.ctypes.data,
        int(tz.size), float(w[0]), float(w[1]), float(w[2]), _p(result), _p(ws),
        int(ws.numel()), stream_ptr())
    if rc == -2:
        return None
    _lib.check(rc, "nsol_corr3_wrap_norms")
    _wrote(out)
    return result


class LanczosBoard(object):
    """Device scalars of an LSMR solve run as Lanczos on the normal equations with both
    halves of a step taken by the blur (nsol_corr3_wrap_lanczos_*): the sums of every
    step (board: |y_j|^2, |A y_j|^2, |grad y_j|^2 at 3 j ..) and the coefficients the
    next kernel reads (coef).  The host reads the board when it needs the scalars --
    the kernels never wait for it."""

    def __init__(self, like, steps, rho_grad, rho_ident):
        self.steps = int(steps)
        self.rho_grad, self.rho_ident = float(rho_grad), float(rho_ident)
        self.board = torch.zeros(3 * (self.steps + 2), dtype=torch.float64,
                                 device=like.device)
        self.coef = torch.zeros(8, dtype=like.dtype, device=like.device)

    def slot_norm2(self, j):
        """The one-element view |y_j|^2 lands in (a reduction kernel's `result`)."""
        return self.board[3 * j:3 * j + 1]

    def init(self):
        _lib.check(_fn("corr3_wrap_lanczos_init", self.coef)(
            _p(self.board), _p(self.coef), self.rho_grad, self.rho_ident, stream_ptr()),
            "nsol_corr3_wrap_lanczos_init")


def corr3_lanczos_a(y, y_prev, t, q0, shape, taps_z, taps_y, taps_x, lb, step):
    """First half of Lanczos step `step` (see LanczosBoard): t = blur(y), q0 = c1 K'K y +
    c0 y + c2 y_prev, sums onto the board.  False when the kernel does not apply
    (nothing launched)."""
    _same(y, t, q0)
    if y_prev is not None:
        _same(y, y_prev)
    ndim, nz, ny, nx = dims3(shape)
    tz, ty, tx = (np.ascontiguousarray(v, dtype=np.float64)
                  for v in (taps_z, taps_y, taps_x))
    if ndim != 3 or nz * ny * nx != y.numel() or not (tz.size == ty.size == tx.size):
        return False
    ws, _ = _workspace(y.device)
    rc = _fn("corr3_wrap_lanczos_a", y)(
        _p(y), _p(y_prev), _p(t), _p(q0), nz, ny, nx, tz.ctypes.data, ty.ctypes.data,
        tx.ctypes.data, int(tz.size), lb.rho_grad, lb.rho_ident, _p(lb.board), int(step),
        _p(lb.coef), _p(ws), int(ws.numel()), stream_ptr())
    if rc == -2:
        return False
    _lib.check(rc, "nsol_corr3_wrap_lanczos_a")
    _wrote(t, q0)
    return True


# True: the halves of a Lanczos step as nsol_corr3_wrap_lanczos_a2 / _b2 (no q0 array:
# the second half forms the step's K'K y itself; 25 B per voxel and step instead of 33)
LEAN_LANCZOS_HALVES = True


def corr3_lanczos_a2(y, t, shape, taps_z, taps_y, taps_x, lb, step):
    """First half, lean form: t = blur(y), its two sums onto the board, the second
    half's coefficients.  False when the kernel does not apply (nothing launched)."""
    _same(y, t)
    ndim, nz, ny, nx = dims3(shape)
    tz, ty, tx = (np.ascontiguousarray(v, dtype=np.float64)
                  for v in (taps_z, taps_y, taps_x))
    if ndim != 3 or nz * ny * nx != y.numel() or not (tz.size == ty.size == tx.size):
        return False
    ws, _ = _workspace(y.device)
    rc = _fn("corr3_wrap_lanczos_a2", y)(
        _p(y), _p(t), nz, ny, nx, tz.ctypes.data, ty.ctypes.data, tx.ctypes.data,
        int(tz.size), lb.rho_grad, lb.rho_ident, _p(lb.board), int(step), _p(lb.coef),
        _p(ws), int(ws.numel()), stream_ptr())
    if rc == -2:
        return False
    _lib.check(rc, "nsol_corr3_wrap_lanczos_a2")
    _wrote(t)
    return True


def corr3_lanczos_b2(t, y, y_prev, y_new, shape, taps_z, taps_y, taps_x, lb, step):
    """Second half, lean form: y_new = ca blur(t) + (c1 K'K y + c0 y + c2 y_prev) + cy y
    with |y_new|^2 onto the board (y_prev may be None; y_new None: the sum alone)."""
    _same(t, y)
    if y_new is not None:
        _same(t, y_new)
    if y_prev is not None:
        _same(y, y_prev)
    ndim, nz, ny, nx = dims3(shape)
    tz, ty, tx = (np.ascontiguousarray(v, dtype=np.float64)
                  for v in (taps_z, taps_y, taps_x))
    if ndim != 3 or nz * ny * nx != y.numel() or not (tz.size == ty.size == tx.size):
        return False
    ws, _ = _workspace(y.device)
    rc = _fn("corr3_wrap_lanczos_b2", y)(
        _p(t), _p(y), _p(y_prev), _p(y_new), nz, ny, nx, tz.ctypes.data, ty.ctypes.data,
        tx.ctypes.data, int(tz.size), lb.rho_grad, lb.rho_ident, _p(lb.board), int(step),
        _p(lb.coef), _p(ws), int(ws.numel()), stream_ptr())
    if rc == -2:
        return False
    _lib.check(rc, "nsol_corr3_wrap_lanczos_b2")
    _wrote(y_new)
    return True


def corr3_wrap_loss(x, b, shape, taps_z, taps_y, taps_x, loss, f_scale, result):
    """rho'(r^2) r for r = blur(x) - b with 1/2 sum rho(r^2) in result[0] (the caller's
    device slot), from the blur itself; None when that kernel does not apply (nothing
    launched): then corr3_wrap and loss_cost_grad(minus=b)."""
    _same(x, b)
    ndim, nz, ny, nx = dims3(shape)
    tz, ty, tx = (np.ascontiguousarray(v, dtype=np.float64)
                  for v in (taps_z, taps_y, taps_x))
    if ndim != 3 or nz * ny * nx != x.numel() or not (tz.size == ty.size == tx.size):
        return None
    ws, _ = _workspace(x.device)
    g = empty_like(x)
    rc = _fn("corr3_wrap_loss", x)(
        _p(x), _p(b), _p(g), nz, ny, nx, tz.ctypes.data, ty.ctypes.data, tx.ctypes.data,
        int(tz.size), LOSSES[loss], float(f_scale), _p(result), _p(ws), int(ws.numel()),
        stream_ptr())
    if rc == -2:
        return None
    _lib.check(rc, "nsol_corr3_wrap_loss")
    return g


def corr3_lanczos_b(t, q0, y, y_new, shape, taps_z, taps_y, taps_x, lb, step):
    """Second half: y_new = ca blur(t) + q0 + cy y with |y_new|^2 onto the board."""
    _same(t, q0, y, y_new)
    ndim, nz, ny, nx = dims3(shape)
    tz, ty, tx = (np.ascontiguousarray(v, dtype=np.float64)
                  for v in (taps_z, taps_y, taps_x))
    if ndim != 3 or nz * ny * nx != y.numel() or not (tz.size == ty.size == tx.size):
        return False
    ws, _ = _workspace(y.device)
    rc = _fn("corr3_wrap_lanczos_b", y)(
        _p(t), _p(q0), _p(y), _p(y_new), nz, ny, nx, tz.ctypes.data, ty.ctypes.data,
        tx.ctypes.data, int(tz.size), lb.rho_grad, lb.rho_ident, _p(lb.board), int(step),
        _p(lb.coef), _p(ws), int(ws.numel()), stream_ptr())
    if rc == -2:
        return False
    _lib.check(rc, "nsol_corr3_wrap_lanczos_b")
    _wrote(y_new)
    return True


def corr_dense(x, shape, taps_dev, kshape3, centre3, mode):
    _chk(x)
    _chk(taps_dev)
    _, nz, ny, nx = dims3(shape)
    out = empty_like(x)
    _lib.check(_fn("corr_dense", x)(
        _p(x), _p(out), nz, ny, nx, _p(taps_dev), kshape3[0], kshape3[1],
        kshape3[2], centre3[0], centre3[1], centre3[2], MODES[mode],
        stream_ptr()), "nsol_corr_dense")
    return _wrote(out)


# --------------------------------------------------------- element-wise ----
def lincomb2(a, x, b, y, out=None):
    _same(x, y)
    if out is None:
        out = empty_like(x)
    else:
        _same(x, out)
    _lib.check(_fn("lincomb2", x)(_p(out), float(a), _p(x), float(b), _p(y),
                                  x.numel(), stream_ptr()), "nsol_lincomb2")
    return _wrote(out)


def lincomb3(a, x, b, y, c, z, out=None):
    _same(x, y, z)
    if out is None:
        out = empty_like(x)
    else:
        _same(x, out)
    _lib.check(_fn("lincomb3", x)(_p(out), float(a), _p(x), float(b), _p(y),
                                  float(c), _p(z), x.numel(), stream_ptr()),
               "nsol_lincomb3")
    return _wrote(out)


def lincomb_many(vecs, coefs, out=None, bounds=None):
    """sum_k coefs[k] * vecs[k] in one pass (up to 40 vectors; the terms are added
    in order): nsol_lb_wcomb_* without base vectors or mask.  bounds = (lo, hi): the
    sum clipped to them in the same pass (nsol_lincomb_clip_*)."""
    import ctypes
    x = _same(vecs[0], *vecs[1:])
    if len(vecs) != len(coefs) or len(vecs) > 40:
        raise ValueError("lincomb_many: %d vectors, %d coefficients"
                         % (len(vecs), len(coefs)))
    if out is None:
        out = empty_like(x)
    else:
        _same(x, out)
    ptrs = (ctypes.c_void_p * len(vecs))(*[v.data_ptr() for v in vecs])
    co = np.ascontiguousarray(coefs, dtype=np.float64)
    if bounds is not None:
        lo, hi = _clip_bounds(x, bounds[0], bounds[1])       # (as clip() has them)
        if lo <= hi:
            _lib.check(_fn("lincomb_clip", x)(
                _p(out), x.numel(), len(vecs), ctypes.cast(ptrs, ctypes.c_void_p),
                co.ctypes.data, lo, hi, stream_ptr()), "nsol_lincomb_clip")
            return _wrote(out)
        return clip(lincomb_many(vecs, coefs, out=out), bounds[0], bounds[1], out=out)
    fn = getattr(_lib.load(), "nsol_lb_wcomb_%s" % suffix(x))
    kt = _timing.active()
    if kt is not None:
        fn = kt.wrap("lb_wcomb", fn)
    _lib.check(fn(_p(out), x.numel(), None, 1.0, 0, None, None, len(vecs),
                  ctypes.cast(ptrs, ctypes.c_void_p), co.ctypes.data,
                  stream_ptr()), "nsol_lb_wcomb")
    return _wrote(out)


def scale(x, a, divide=False, out=None):
    _chk(x)
    if out is None:
        out = empty_like(x)
    _lib.check(_fn("scale", x)(_p(out), _p(x), float(a), int(bool(divide)),
                               x.numel(), stream_ptr()), "nsol_scale")
    return _wrote(out)


def _clip_bounds(x, lo, hi):
    lo = -1.7976931348623157e308 if lo == -np.inf else float(lo)
    hi = 1.7976931348623157e308 if hi == np.inf else float(hi)
    if x.dtype == torch.float32:
        lo = max(lo, -3.4028234663852886e38)
        hi = min(hi, 3.4028234663852886e38)
    return lo, hi


def clip(x, lo, hi, out=None):
    _chk(x)
    if out is None:
        out = empty_like(x)
    lo, hi = _clip_bounds(x, lo, hi)
    _lib.check(_fn("clip", x)(_p(out), _p(x), lo, hi, x.numel(),
                              stream_ptr()), "nsol_clip")
    return _wrote(out)


def prox_dual_clamp(x, den=1.0, out=None):
    _chk(x)
    if out is None:
        out = empty_like(x)
    _lib.check(_fn("prox_dual_clamp", x)(_p(out), _p(x), float(den),
                                         x.numel(), stream_ptr()),
               "nsol_prox_dual_clamp")
    return _wrote(out)


def prox_ell2(x, bt, tau, out=None):
    _same(x, bt)
    if out is None:
        out = empty_like(x)
    _lib.check(_fn("prox_ell2", x)(_p(out), _p(x), _p(bt), float(tau),
                                   x.numel(), stream_ptr()), "nsol_prox_ell2")
    return _wrote(out)


def prox_ell1(x, bt, tau, out=None):
    _same(x, bt)
    if out is None:
        out = empty_like(x)
    _lib.check(_fn("prox_ell1", x)(_p(out), _p(x), _p(bt), float(tau),
                                   x.numel(), stream_ptr()), "nsol_prox_ell1")
    return _wrote(out)


# ------------------------------------------------------------ reductions ----
_ws = {}


def _workspace(dev):
    key = (dev.index, stream_ptr())
    if key not in _ws:
        n = _lib.load().nsol_hip_reduce_ws_doubles()
        _ws[key] = (torch.empty(n, dtype=torch.float64, device=dev),
                    torch.empty(1, dtype=torch.float64, device=dev))
    return _ws[key]


def dot(x, y):
    """sum(x*y) accumulated in float64; returns a Python float (syncs)."""
    _same(x, y)
    ws, res = _workspace(x.device)
    _lib.check(_fn("dot", x)(_p(x), _p(y), x.numel(), _p(res), _p(ws),
                             stream_ptr()), "nsol_dot")
    return float(res.item())


def norm2(x):
    return float(np.sqrt(dot(x, x)))


def loss_cost_grad(r, loss, f_scale, want_grad=True, out=None, minus=None,
                   result=None):
    """(0.5*sum rho(r^2), rho'(r^2)*r); with `minus` the residual is r - minus,
    formed in the same pass.  result: the caller's one-element float64 device
    tensor for the cost (then nothing is read back and it is returned as it is)."""
    _chk(r)
    ws, res = _workspace(r.device)
    if result is not None:
        res = result
    g = None
    if want_grad:
        g = empty_like(r) if out is None else out
    if minus is None:
        _lib.check(_fn("loss_cost_grad", r)(
            _p(r), _p(g), r.numel(), LOSSES[loss], float(f_scale), _p(res),
            _p(ws), stream_ptr()), "nsol_loss_cost_grad")
    else:
        _same(r, minus)
        _lib.check(_fn("loss_residual_cost_grad", r)(
            _p(r), _p(minus), _p(g), r.numel(), LOSSES[loss], float(f_scale),
            _p(res), _p(ws), stream_ptr()), "nsol_loss_residual_cost_grad")
    return (res if result is not None else float(res.item())), _wrote(g)


def tk1_reg_cost_grad(x, g, shape, w, alpha, out=None, result=None):
    """(sum |grad x|^2, g + alpha * grad_adj(grad x)) in one pass over x
    (the regulariser's share of tikhonov_linear_solver.py:201-208 with
    B = gradient).  out may be g.  result: as in loss_cost_grad."""
    _same(x, g)
    ndim, nz, ny, nx = dims3(shape)
    if out is None:
        out = empty_like(g)
    ws, res = _workspace(x.device)
    if result is not None:
        res = result
    _lib.check(_fn("tk1_reg_cost_grad", x)(
        _p(x), _p(g), _p(out), ndim, nz, ny, nx, w[0], w[1], w[2], float(alpha),
        _p(res), _p(ws), stream_ptr()), "nsol_tk1_reg_cost_grad")
    return (res if result is not None else float(res.item())), _wrote(out)


_ws3 = {}


def tk1_reg_objective(x, g, d, shape, w, alpha, lo, hi, out, result, gold=None,
                      ydiff=None):
    """tk1_reg_cost_grad with the new gradient's product with d (None: 0) and its
    largest projected component for lo <= x <= hi in result[1], result[2]; with gold,
    ydiff = out - gold and its sum of squares in result[3] (result: the caller's
    four-element float64 device slots; nothing is read back here)."""
    _same(x, g)
    if d is not None:
        _same(x, d)
    if gold is not None:
        _same(x, gold, ydiff)
    ndim, nz, ny, nx = dims3(shape)
    key = (x.device.index, stream_ptr())
    if key not in _ws3:
        _ws3[key] = torch.empty(4 * _lib.load().nsol_hip_reduce_ws_doubles(),
                                dtype=torch.float64, device=x.device)
    _lib.check(_fn("tk1_reg_objective", x)(
        _p(x), _p(g), _p(out), _p(d), _p(gold), _p(ydiff), ndim, nz, ny, nx, w[0], w[1],
        w[2], float(alpha), float(lo), float(hi), _p(result), _p(_ws3[key]),
        stream_ptr()), "nsol_tk1_reg_objective")
    return _wrote(out, ydiff)


def tk1_grad_norm(x, shape, w, result=None):
    """sum |grad x|^2 alone (one read of x); result: the caller's device slot."""
    _chk(x)
    ndim, nz, ny, nx = dims3(shape)
    ws, res = _workspace(x.device)
    if result is not None:
        res = result
    _lib.check(_fn("tk1_grad_norm", x)(
        _p(x), ndim, nz, ny, nx, w[0], w[1], w[2], _p(res), _p(ws), stream_ptr()),
        "nsol_tk1_grad_norm")
    return res if result is not None else float(res.item())


def tk1_lanczos(x, g, z, shape, w, alpha, c_g, c_x, c_z, out, result=None):
    """out = c_g g + alpha grad_adj(grad x) + c_x x + c_z z (z may be None) with the
    sum of squares of out: the Lanczos update of the normal equations."""
    _same(x, g, out)
    if z is not None:
        _same(x, z)
    ndim, nz, ny, nx = dims3(shape)
    ws, res = _workspace(x.device)
    if result is not None:
        res = result
    _lib.check(_fn("tk1_lanczos", x)(
        _p(x), _p(g), _p(z), _p(out), ndim, nz, ny, nx, w[0], w[1], w[2],
        float(alpha), float(c_g), float(c_x), float(c_z), _p(res), _p(ws),
        stream_ptr()), "nsol_tk1_lanczos")
    _wrote(out)
    return res if result is not None else float(res.item())


class ScalarFetch(object):
    """Device scalars to the host while later kernels run: an event behind the
    kernels that produce them, a copy into pinned memory on a side stream, and a
    wait on that copy alone -- the stream the solver enqueues on is not drained
    (a `.cpu()` on it would wait for everything enqueued so far)."""

    def __init__(self, device, count):
        self.host = torch.empty(count, dtype=torch.float64).pin_memory()
        self.side = torch.cuda.Stream(device=device)
        self.ready = torch.cuda.Event()
        self.done = torch.cuda.Event()

    def start(self, dev_scalars):
        self.ready.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ready)
            self.host[:dev_scalars.numel()].copy_(dev_scalars, non_blocking=True)
            self.done.record(self.side)

    def wait(self):
        self.done.synchronize()
        return self.host.numpy()


_fetchers = {}


def scalar_fetchers(device, count, n):
    """n ScalarFetch objects of `count` doubles each, kept per device and stream (pinned
    memory and a side stream cost far more to create than to use)."""
    key = (device.index, stream_ptr(), int(count), int(n))
    if key not in _fetchers:
        _fetchers[key] = [ScalarFetch(device, count) for _ in range(n)]
    return _fetchers[key]


_ws8 = {}


def pair_stats(x, y, mx=0.0, my=0.0):
    """NumPy array of the 8 sums documented at nsol_pair_stats_* (syncs)."""
    _same(x, y)
    ws, _ = _workspace(x.device)
    key = (x.device.index, stream_ptr())
    if key not in _ws8:
        _ws8[key] = torch.empty(8, dtype=torch.float64, device=x.device)
    res = _ws8[key]
    _lib.check(_fn("pair_stats", x)(_p(x), _p(y), x.numel(), float(mx),
                                    float(my), _p(res), _p(ws), stream_ptr()),
               "nsol_pair_stats")
    return res.cpu().numpy()


def loss_eval(f2, loss, f_scale=1.0, huber_gamma=1.345):
    """Element-wise (rho(f2), rho'(f2))."""
    _chk(f2)
    rho, drho = empty_like(f2), empty_like(f2)
    _lib.check(_fn("loss_eval", f2)(_p(f2), _p(rho), _p(drho), f2.numel(),
                                    LOSSES[loss], float(f_scale),
                                    float(huber_gamma), stream_ptr()),
               "nsol_loss_eval")
    return rho, drho


def vector_norm_sum(t, ndim, mode=0, gamma=0.05):
    _chk(t)
    ws, res = _workspace(t.device)
    _lib.check(_fn("vector_norm_sum", t)(
        _p(t), int(ndim), t.numel() // int(ndim), int(mode), float(gamma),
        _p(res), _p(ws), stream_ptr()), "nsol_vector_norm_sum")
    return float(res.item())


# ------------------------------------------------------------------- PD ----
def pd_dual_step(xbar, p_in, p_out, shape, w, sigma, hden):
    ndim, nz, ny, nx = dims3(shape)
    _lib.check(_fn("pd_dual_step", xbar)(
        _p(xbar), _p(p_in), _p(p_out), ndim, nz, ny, nx, w[0], w[1], w[2],
        float(sigma), float(hden), stream_ptr()), "nsol_pd_dual_step")
    _wrote(p_out)


def pd_primal_step(p, x, xbar, bt, shape, w, tau, tl, theta, flags):
    ndim, nz, ny, nx = dims3(shape)
    _lib.check(_fn("pd_primal_step", x)(
        _p(p), _p(x), _p(xbar), _p(bt), ndim, nz, ny, nx, w[0], w[1], w[2],
        float(tau), float(tl), float(theta), int(flags), stream_ptr()),
        "nsol_pd_primal_step")
    _wrote(x, xbar)


def pd_fused_iter(xbar_in, xbar_out, x, bt, p_in, p_out, shape, w, sigma,
                  hden, tau, tl, theta, flags):
    ndim, nz, ny, nx = dims3(shape)
    _lib.check(_fn("pd_fused_iter", x)(
        _p(xbar_in), _p(xbar_out), _p(x), _p(bt), _p(p_in), _p(p_out), ndim,
        nz, ny, nx, w[0], w[1], w[2], float(sigma), float(hden), float(tau),
        float(tl), float(theta), int(flags), stream_ptr()),
        "nsol_pd_fused_iter")
    _wrote(xbar_out, x, p_out)


def pd_fused2_iter(xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out, shape, w,
                   sigma2, hden2, tau2, tl2, theta2, flags):
    """Two iterations in one pass; returns False if the kernel does not apply
    to this problem (nothing was launched)."""
    ndim, nz, ny, nx = dims3(shape)
    arr = [np.ascontiguousarray(a, dtype=np.float64)
           for a in (sigma2, hden2, tau2, tl2, theta2)]
    rc = _fn("pd_fused2_iter", x_in)(
        _p(xbar_in), _p(xbar_out), _p(x_in), _p(x_out), _p(bt), _p(p_in),
        _p(p_out), ndim, nz, ny, nx, w[0], w[1], w[2], arr[0].ctypes.data,
        arr[1].ctypes.data, arr[2].ctypes.data, arr[3].ctypes.data,
        arr[4].ctypes.data, int(flags), stream_ptr())
    if rc == -2:
        return False
    _lib.check(rc, "nsol_pd_fused2_iter")
    _wrote(xbar_out, x_out, p_out)
    return True


def pd_fusedk_iter(xbar_in, xbar_out, x_in, x_out, bt, p_in, p_out, shape, w,
                   sigma, hden, tau, tl, theta, flags):
    """len(sigma) = 2 or 3 iterations in one pass on tiled footprints; returns
    False if the kernel does not apply (nothing was launched)."""
    ndim, nz, ny, nx = dims3(shape)
    arr = [np.ascontiguousarray(a, dtype=np.float64)
           for a in (sigma, hden, tau, tl, theta)]
    k = int(arr[0].size)
    if any(a.size != k for a in arr):
        raise ValueError("step-size arrays must have equal length")
    rc = _fn("pd_fusedk_iter", x_in)(
        _p(xbar_in), _p(xbar_out), _p(x_in), _p(x_out), _p(bt), _p(p_in),
        _p(p_out), ndim, nz, ny, nx, w[0], w[1], w[2], k, arr[0].ctypes.data,
        arr[1].ctypes.data, arr[2].ctypes.data, arr[3].ctypes.data,
        arr[4].ctypes.data, int(flags), stream_ptr())
    if rc == -2:
        return False
    _lib.check(rc, "nsol_pd_fusedk_iter")
    _wrote(xbar_out, x_out, p_out)
    return True


def pd_fusedk_tuned(x, shape, k=3):
    """1 once the online tuner of the k-iterations-per-pass kernel has settled
    for this shape / dtype, 0 while exploring, -1 if the shape is unknown."""
    ndim, nz, ny, nx = dims3(shape)
    return int(_lib.load().nsol_pd_fusedk_tuned(int(x.element_size()), int(k),
                                                nz, ny, nx))


def pd_fusedk_launches(k=3):
    """k_pd_fusedk launches of depth k made by this process so far."""
    return int(_lib.load().nsol_pd_fusedk_launches(int(k)))


def pd_fusedk_plan(x, shape, k=3):
    """(waves, tiles along x, z-chunk) the online tuner settled on, or None."""
    import ctypes
    ndim, nz, ny, nx = dims3(shape)
    wv, nt, zc = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
    rc = _lib.load().nsol_pd_fusedk_plan(
        int(x.element_size()), int(k), nz, ny, nx, ctypes.byref(wv),
        ctypes.byref(nt), ctypes.byref(zc))
    return None if rc else (int(wv.value), int(nt.value), int(zc.value))


# Cache-resident volumes (BASELINE configs 1-2): the whole run in ONE launch of the
# persistent kernel (nsol_pd_persist_run_*), from this many iterations on and up to
# this many voxels (above, three iterations per pass take over).
PD_PERSIST = True
PD_PERSIST_MIN_ITERS = 16
PD_PERSIST_MAX_VOXELS = 1 << 20
_persist_launches = 0


def pd_persist_launches():
    """Runs that went through the persistent kernel (for tests and tools)."""
    return _persist_launches


# Error words of the persistent runs: a ring of 32-bit words in pinned host
# memory the kernels write to directly.  A persistent run reads its state from
# one set of arrays and writes the result to another, so a run whose word came
# back raised (a workgroup gave up waiting for a neighbour: the device is shared
# or CU-masked and not every workgroup was resident) is REPEATED from the
# untouched inputs with one launch per iteration -- the same bits -- by
# settle_persist_runs().  That is called wherever a result is consumed after a
# synchronisation the caller performs anyway: Solver.run() (after its own
# torch.cuda.synchronize), device.to_numpy, solve_batch before its gather, the
# next pd_run on the device; never with a synchronisation in the middle of
# enqueued work.
_ERR_SLOTS = 256
_err_ring = None
_err_next = 0
_pending_runs = []         # PersistRun records, oldest first
_fallback_warned = False
persist_fallbacks = 0      # runs repeated so far (tests, tools)


class PersistRun(object):
    """What it takes to repeat a persistent run through nsol_pd_run_*."""

    def __init__(self, slot, stream, src, dst, bt, shape, w, lmbda, sigma, tau,
                 theta, p_is_zero, gamma_huber, flags, keep):
        self.slot, self.stream = slot, stream
        self.src, self.dst = src, dst         # (xbar, x, p) tensors read / written
        self.bt, self.shape, self.w, self.lmbda = bt, shape, w, lmbda
        self.sigma, self.tau, self.theta = sigma, tau, theta
        self.p_is_zero, self.gamma_huber, self.flags = p_is_zero, gamma_huber, flags
        self.keep = keep                      # tensors that must outlive the run


def _err_slot():
    global _err_ring, _err_next
    if _err_ring is None:
        _err_ring = torch.zeros(_ERR_SLOTS, dtype=torch.int32).pin_memory()
    slot = _err_next
    _err_next = (_err_next + 1) % _ERR_SLOTS
    if any(r.slot == slot for r in _pending_runs):
        settle_persist_runs()             # the ring has wrapped: settle first
    _err_ring[slot] = 0
    return slot


def _repeat_with_a_launch_per_iteration(r):
    """The inputs of run r are intact (it wrote elsewhere): the same iterations
    through nsol_pd_run_*, the result put where the persistent run left it."""
    import ctypes
    xb_in, x_in, p_in = r.src
    xb_out, x_out, p_out = r.dst
    ndim, nz, ny, nx = dims3(r.shape)
    n_it = int(r.sigma.size)
    slot = ctypes.c_int(0)
    # scratch state so that neither the inputs' nor the outputs' roles matter:
    # start from copies of the inputs, ping-pong against the output arrays
    xb0, x0, p0 = xb_in.clone(), x_in.clone(), p_in.clone()
    _lib.check(_fn("pd_run", x0)(
        _p(xb0), _p(xb_out), _p(x0), _p(x_out), _p(r.bt), _p(p0), _p(p_out), ndim,
        nz, ny, nx, r.w[0], r.w[1], r.w[2], float(r.lmbda), r.sigma.ctypes.data,
        r.tau.ctypes.data, r.theta.ctypes.data, n_it, int(bool(r.p_is_zero)),
        float(r.gamma_huber), int(r.flags) | PD_RUN_X_MAY_SWAP,
        ctypes.addressof(slot), stream_ptr()), "nsol_pd_run")
    _wrote(xb_out, x_out, p_out)
    if not (slot.value & 1):              # final xbar / p in the scratch pair
        xb_out.copy_(xb0)
        p_out.copy_(p0)
    if not (slot.value & 2):              # final x in x0, not in x_out
        x_out.copy_(x0)
    torch.cuda.synchronize()


def settle_persist_runs(synchronize=True):
    """Look at the error word of every persistent run enqueued so far; repeat
    the runs that timed out (see above); returns how many were repeated.
    synchronize=False: the caller has just synchronised the device."""
    global _fallback_warned, persist_fallbacks
    if not _pending_runs:
        return 0
    if synchronize:
        torch.cuda.synchronize()
    saved = (globals()["PD_PERSIST"],)
    repeated = 0
    while _pending_runs:
        r = _pending_runs.pop(0)
        if int(_err_ring[r.slot]) == 0:
            continue
        if r.dst[0] is r.src[0]:          # in place: the inputs are gone
            del _pending_runs[:]
            raise _lib.NsolHipError(
                "nsol_pd_persist_run: a workgroup gave up waiting for a neighbour "
                "(device shared or CU-masked?) in a run that updated its state in "
                "place; the result is invalid")
        if not _fallback_warned:
            import warnings
            warnings.warn(
                "nsol_pd_persist_run: a workgroup gave up waiting for a "
                "neighbour (device shared or CU-masked?); the run is repeated "
                "with one launch per iteration -- set nsol_amd.ops.PD_PERSIST "
                "= False to skip the attempt", RuntimeWarning)
            _fallback_warned = True
        persist_fallbacks += 1
        repeated += 1
        globals()["PD_PERSIST"] = False
        try:
            _repeat_with_a_launch_per_iteration(r)
        finally:
            globals()["PD_PERSIST"] = saved[0]
    return repeated


def drain_persist_checks():
    """(former name) settle after the caller's own synchronisation."""
    settle_persist_runs(synchronize=False)


def pd_persist_run(xbar, x, bt, p, shape, w, lmbda, sigma, tau, theta, p_is_zero,
                   gamma_huber, flags, out=None):
    """All len(sigma) iterations in one launch.  out = (xbar_out, x_out, p_out):
    where the result goes -- arrays other than the inputs, which then survive a
    timed-out run (settle_persist_runs repeats it); out=None: in place, and a
    time-out surfaces as an exception from settle_persist_runs instead.  Returns
    False when the kernel does not apply (nothing launched).  Does not
    synchronise."""
    global _persist_launches
    ndim, nz, ny, nx = dims3(shape)
    iters = int(np.size(sigma))
    need = int(_lib.load().nsol_pd_persist_ws_bytes(int(x.element_size()), ndim,
                                                    nz, ny, nx, iters))
    if need < 0:
        return False
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    slot = _err_slot()
    dst = (xbar, x, p) if out is None else tuple(out)
    rc = _fn("pd_persist_run_to", x)(
        _p(xbar), _p(x), _p(bt), _p(p), _p(dst[0]), _p(dst[1]), _p(dst[2]), ndim,
        nz, ny, nx, w[0], w[1], w[2],
        float(lmbda), sigma.ctypes.data, tau.ctypes.data, theta.ctypes.data, iters,
        int(bool(p_is_zero)), float(gamma_huber), int(flags), _p(ws), need,
        _err_ring.data_ptr() + 4 * slot, stream_ptr())
    if rc == -2:
        return False
    _lib.check(rc, "nsol_pd_persist_run")
    _wrote(*dst)
    ws.record_stream(torch.cuda.current_stream())      # freed once the run is done
    _pending_runs.append(PersistRun(
        slot, stream_ptr(), (xbar, x, p), dst, bt, tuple(shape), tuple(w), lmbda,
        sigma, tau, theta, p_is_zero, gamma_huber, flags, keep=(ws,)))
    _persist_launches += 1
    return True


def persist_pays(shape, iterations):
    """Where one launch per run beats one launch per iteration (tools/
    bench_persist.py): the persistent kernel takes 4.2-4.8 us per iteration whatever
    the size (the hand-off between workgroups) plus ~20 us per run; a launch per
    iteration costs 5.3-6.7 us on 3-D volumes but only 3.4 us on small images."""
    n = int(np.prod(shape))
    if n > PD_PERSIST_MAX_VOXELS or iterations < PD_PERSIST_MIN_ITERS:
        return False
    return len(shape) == 3 or n >= (1 << 19)


PD_RUN_X_MAY_SWAP = 0x100


def row_pitch(shape, like):
    """Elements between the starts of consecutive rows of a padded 3-D layout: the row
    length rounded up to whole 16-byte vectors (0 where the rows are whole already, or
    the volume is not 3-D)."""
    if len(tuple(shape)) != 3:
        return 0
    vec = 16 // like.element_size()
    nx = int(shape[2])
    if nx < 2 * vec:
        # (the pitched one-iteration kernel -- a run's odd trailing iteration -- needs
        # two vectors per row, nsol_pd.hip fused_iter_impl: such rows stay contiguous)
        return 0
    return 0 if nx % vec == 0 else (nx + vec - 1) // vec * vec


def to_pitched(t, shape, pitch, comps=1, fill=0.0):
    """A flat (comps x volume) tensor re-laid with its rows at `pitch` (padding = fill)."""
    nz, ny, nx = (int(v) for v in shape)
    out = torch.full((comps * nz * ny * pitch,), fill, dtype=t.dtype, device=t.device)
    out.view(comps * nz, ny, pitch)[:, :, :nx].copy_(t.view(comps * nz, ny, nx))
    return _wrote(out)


def from_pitched(t, shape, pitch, comps=1):
    """The contiguous tensor back out of a pitched one."""
    nz, ny, nx = (int(v) for v in shape)
    return t.view(comps * nz, ny, pitch)[:, :, :nx].contiguous().view(-1)


def pd_run(xbar0, xbar1, x, bt, p0, p1, shape, w, lmbda, sigma, tau, theta,
           p_is_zero, gamma_huber, flags, x_alt=None, swap_ok=False, pitch=0):
    """Enqueue len(sigma) iterations; returns the slot (0/1) of xbar/p that
    holds the final state.  x holds the final primal iterate.

    pitch > 0: every array is held with its rows at that pitch (row_pitch /
    to_pitched; nsol_pd_run_pitched_*).

    swap_ok: x and x_alt are whole tensors nobody else aliases -- when the
    multi-iteration kernels leave the final iterate in x_alt, the two tensors
    trade their storage (x still names the result) instead of a copy of the
    volume."""
    import ctypes
    ndim, nz, ny, nx = dims3(shape)
    if _pending_runs:
        # an earlier persistent run may feed this one: its verdict first (the
        # device is drained here; runs of cache-resident volumes are ~1 ms)
        settle_persist_runs()
    if pitch:
        sigma = np.ascontiguousarray(sigma, dtype=np.float64)
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        slot = ctypes.c_int(0)
        _lib.check(_fn("pd_run_pitched", x)(
            _p(xbar0), _p(xbar1), _p(x), _p(x_alt), _p(bt), _p(p0), _p(p1), ndim,
            nz, ny, nx, int(pitch), w[0], w[1], w[2], float(lmbda), sigma.ctypes.data,
            tau.ctypes.data, theta.ctypes.data, int(sigma.size), int(bool(p_is_zero)),
            float(gamma_huber),
            int(flags) | (PD_RUN_X_MAY_SWAP if swap_ok and x_alt is not None else 0),
            ctypes.addressof(slot), stream_ptr()), "nsol_pd_run_pitched")
        _wrote(xbar0, xbar1, x, x_alt, p0, p1)
        if slot.value & 2:
            x.data, x_alt.data = x_alt.data, x.data
        return int(slot.value) & 1
    if PD_PERSIST and persist_pays(shape, np.size(sigma)):
        # the result goes to the other half of the ping-pong arrays (and to the
        # x scratch volume): the inputs stay as they are until the run's error
        # word has been looked at
        xo = x_alt if x_alt is not None else empty_like(x)
        if pd_persist_run(xbar0, x, bt, p0, shape, w, lmbda, sigma, tau, theta,
                          p_is_zero, gamma_huber, flags, out=(xbar1, xo, p1)):
            if swap_ok and x_alt is not None:
                x.data, x_alt.data = x_alt.data, x.data
                # (the record holds tensors, not storages: keep it pointing at
                # the input / output storages after the swap of names)
                r = _pending_runs[-1]
                r.src = (r.src[0], x_alt, r.src[2])
                r.dst = (r.dst[0], x, r.dst[2])
            else:
                # x must hold the result: the input x survives in a copy
                r = _pending_runs[-1]
                keep = x.clone()
                x.copy_(xo)
                r.src = (r.src[0], keep, r.src[2])
                r.dst = (r.dst[0], x, r.dst[2])
            return 1
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    slot = ctypes.c_int(0)
    _lib.check(_fn("pd_run", x)(
        _p(xbar0), _p(xbar1), _p(x), _p(x_alt), _p(bt), _p(p0), _p(p1), ndim,
        nz, ny, nx, w[0], w[1], w[2], float(lmbda), sigma.ctypes.data,
        tau.ctypes.data, theta.ctypes.data, int(sigma.size),
        int(bool(p_is_zero)), float(gamma_huber),
        int(flags) | (PD_RUN_X_MAY_SWAP if swap_ok and x_alt is not None else 0),
        ctypes.addressof(slot), stream_ptr()), "nsol_pd_run")
    _wrote(xbar0, xbar1, x, x_alt, p0, p1)
    if slot.value & 2:
        x.data, x_alt.data = x_alt.data, x.data
    return int(slot.value) & 1


# ----------------------------------------------------------------- ADMM ----
def admm_vw_update(x, v, w_, c, rhs, shape, w, thr, rhs_scale, want_norm=False):
    """v, w_ and the next right-hand side rhs = rhs_scale * (v - w_ + c) from one
    pass over x; want_norm: returns sum(rhs^2) (syncs)."""
    ndim, nz, ny, nx = dims3(shape)
    if want_norm:
        ws, res = _workspace(x.device)
        _lib.check(_fn("admm_vw_update_norm", x)(
            _p(x), _p(v), _p(w_), _p(c), _p(rhs), ndim, nz, ny, nx, w[0], w[1],
            w[2], float(thr), float(rhs_scale), _p(res), _p(ws), stream_ptr()),
            "nsol_admm_vw_update_norm")
        _wrote(v, w_, rhs)
        return float(res.item())
    _lib.check(_fn("admm_vw_update", x)(
        _p(x), _p(v), _p(w_), _p(c), _p(rhs), ndim, nz, ny, nx, w[0], w[1],
        w[2], float(thr), float(rhs_scale), stream_ptr()),
        "nsol_admm_vw_update")
    _wrote(v, w_, rhs)


def admm_vw_update_g(x, w_in, w_out, c, atb, g, shape, w, thr, rhs_scale, c_atu, c_btu,
                     result):
    """The outer update and the vector the next x-update's LSMR starts from in one pass
    (nsol_admm_vw_update_g_*): w_out, g as admm_vw_update(x, None, w, c, rhs, ...,
    rhs_scale) followed by lsmr_v_update(atb, rhs, atb, B_GRAD, ..., c_atu, c_btu, 0,
    out=g) leave them, rhs never written; result[0] = sum rhs^2, result[1] = sum g^2
    (the caller's two-element float64 device tensor, not read back here).  False when
    the kernel does not apply (nothing launched)."""
    _same(x, atb, g)
    _same(w_in, w_out)
    if c is not None:
        _same(w_in, c)
    ndim, nz, ny, nx = dims3(shape)
    if w_in.dtype != x.dtype or w_in.numel() != ndim * x.numel() or \
            nz * ny * nx != x.numel():
        raise ValueError("admm_vw_update_g: operand sizes do not fit shape %r" %
                         (tuple(shape),))
    _chk(result)
    if result.numel() != 2 or result.dtype != torch.float64:
        raise ValueError("admm_vw_update_g: result must hold two float64 values")
    ws, _ = _workspace(x.device)
    rc = _fn("admm_vw_update_g", x)(
        _p(x), _p(w_in), _p(w_out), _p(c), _p(atb), _p(g), ndim, nz, ny, nx, w[0], w[1],
        w[2], float(thr), float(rhs_scale), float(c_atu), float(c_btu), _p(result), _p(ws),
        stream_ptr())
    if rc == -2:
        return False
    _lib.check(rc, "nsol_admm_vw_update_g")
    _wrote(w_out, g)
    return True


def vector_shrink(t, ndim, thr, out=None):
    _chk(t)
    if out is None:
        out = empty_like(t)
    _lib.check(_fn("vector_shrink", t)(_p(t), _p(out), int(ndim),
                                       t.numel() // int(ndim), float(thr),
                                       stream_ptr()), "nsol_vector_shrink")
    return _wrote(out)


# ----------------------------------------------------------------- LSMR ----
B_NONE, B_GRAD, B_IDENTITY = 0, 1, 2


def flat_geometry(n):
    """(ny, nx) with ny * nx = n for the element-wise modes of the LSMR kernels
    (B = identity / no regulariser): their thread mapping wants rows of at most
    a few thousand elements, a flat vector of 512^3 elements is folded; without
    a suitable power-of-two factor it stays one long row (the stencil mapping
    then loops over x)."""
    n = int(n)
    if n <= (1 << 16):
        return 1, n
    for nx in (4096, 2048, 1024, 512, 256, 128, 64):
        if n % nx == 0:
            return n // nx, nx
    return 1, n


def _bgeom(bmode, shape, w, n):
    if bmode == B_GRAD:
        return dims3(shape), w
    geom = flat_geometry(n)
    if geom is None:
        raise ValueError("no row folding for a vector of %d elements" % n)
    ny, nx = geom
    return ((1 if ny == 1 else 2), 1, ny, nx), (1.0, 1.0, 1.0)


def lsmr_u_update(Av, v, u_top, u_bot, bmode, shape, w, c_av, c_bv, c_u,
                  sync=True, result=None):
    """u_top = c_av*Av + c_u*u_top; u_bot = c_bv*B(v) + c_u*u_bot;
    returns ||[u_top; u_bot]||^2 (sync=False: the device scalar, not read
    back)."""
    if Av is None:               # lower block only (top: corr3_wrap_axpby)
        _same(u_top, v)
        Av_ptr, nn = None, u_top.numel()
    else:
        _same(Av, v, u_top)
        Av_ptr, nn = Av, Av.numel()
    if u_bot is not None:
        _chk(u_bot)
        rows = (len(tuple(shape)) if bmode == B_GRAD else 1) * nn
        if u_bot.dtype != u_top.dtype or u_bot.numel() != rows:
            raise ValueError("lsmr_u_update: lower block has %d elements, "
                             "expected %d" % (u_bot.numel(), rows))
    (ndim, nz, ny, nx), w = _bgeom(bmode, shape, w, nn)
    ws, res = _workspace(u_top.device)
    if result is not None:       # the caller's slot: not read back here
        res, sync = result, False
    _lib.check(_fn("lsmr_u_update", u_top)(
        _p(Av_ptr), _p(v), _p(u_top), _p(u_bot), int(bmode), ndim, nz, ny, nx,
        w[0], w[1], w[2], float(c_av), float(c_bv), float(c_u), _p(res),
        _p(ws), stream_ptr()), "nsol_lsmr_u_update")
    _wrote(u_top, u_bot)
    return float(res.item()) if sync else res


def lsmr_v_update(Atu, u_bot, v, bmode, shape, w, c_atu, c_btu, c_v,
                  sync=True, out=None, result=None):
    """v = c_atu*Atu + c_btu*B^T(u_bot) + c_v*v; returns ||v||^2.  out: where
    the new vector goes instead of v (may be Atu: v then stays as it was)."""
    _same(Atu, v)
    if out is not None:
        _same(Atu, out)
    if u_bot is not None:
        _chk(u_bot)
        rows = (len(tuple(shape)) if bmode == B_GRAD else 1) * Atu.numel()
        if u_bot.dtype != Atu.dtype or u_bot.numel() != rows:
            raise ValueError("lsmr_v_update: lower block has %d elements, "
                             "expected %d" % (u_bot.numel(), rows))
    (ndim, nz, ny, nx), w = _bgeom(bmode, shape, w, Atu.numel())
    ws, res = _workspace(Atu.device)
    if result is not None:       # the caller's slot: not read back here
        res, sync = result, False
    if out is not None:
        _lib.check(_fn("lsmr_v_update_to", Atu)(
            _p(Atu), _p(u_bot), _p(v), _p(out), int(bmode), ndim, nz, ny, nx,
            w[0], w[1], w[2], float(c_atu), float(c_btu), float(c_v), _p(res),
            _p(ws), stream_ptr()), "nsol_lsmr_v_update_to")
        _wrote(out)
        return float(res.item()) if sync else res
    _lib.check(_fn("lsmr_v_update", Atu)(
        _p(Atu), _p(u_bot), _p(v), int(bmode), ndim, nz, ny, nx, w[0], w[1],
        w[2], float(c_atu), float(c_btu), float(c_v), _p(res), _p(ws),
        stream_ptr()), "nsol_lsmr_v_update")
    _wrote(v)
    return float(res.item()) if sync else res


def lsmr_hx_update(hbar, x, h, v, c_hbar, c_x, c_h, c_v, sync=True):
    """hbar = h + c_hbar*hbar; x += c_x*hbar; h = c_v*v + c_h*h;
    returns ||x||^2."""
    _same(x, hbar, h, v)
    ws, res = _workspace(x.device)
    _lib.check(_fn("lsmr_hx_update", x)(
        _p(hbar), _p(x), _p(h), _p(v), x.numel(), float(c_hbar), float(c_x),
        float(c_h), float(c_v), _p(res), _p(ws), stream_ptr()),
        "nsol_lsmr_hx_update")
    _wrote(hbar, x, h)
    return float(res.item()) if sync else res
