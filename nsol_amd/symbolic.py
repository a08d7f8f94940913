"""Symbolic probe used by the solvers to recognise native operator / prox
configurations THROUGH the lambda wrappers that NSoL callers write, e.g.
    D_1D = lambda x: grad(x.reshape(*X_shape)).flatten()       (run_denoising.py:106)
    prox_f = lambda x, tau: prox.prox_ell2_denoising(x, tau, x0=b, x_scale=s)
The solver calls such a callable once with a `Sym`; nsol_amd operators answer a
Sym with a descriptor of themselves, anything else (NumPy arithmetic on the
probe, foreign code) aborts the trace and the solver falls back to the generic
un-fused loop.
"""
import numpy as np


class TraceAbort(Exception):
    pass


class TauSym(float):
    """Sentinel step size handed to prox callables while tracing; arithmetic on
    it yields a plain float, so `isinstance(v, TauSym)` proves pass-through."""


class Sym(object):
    __array_priority__ = 1e6

    def __init__(self, shape, desc=None):
        self._shape = tuple(int(s) for s in shape)
        self.desc = desc

    # ---- array-like surface used by caller-side wrappers
    @property
    def shape(self):
        return self._shape

    @property
    def size(self):
        return int(np.prod(self._shape))

    @property
    def ndim(self):
        return len(self._shape)

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = [int(s) for s in shape]
        if shape.count(-1) == 1:
            known = int(np.prod([s for s in shape if s != -1]))
            shape[shape.index(-1)] = self.size // max(known, 1)
        if int(np.prod(shape)) != self.size:
            raise ValueError("cannot reshape probe of size %d into %s" %
                             (self.size, tuple(shape)))
        return Sym(shape, self.desc)

    def flatten(self):
        return Sym((self.size,), self.desc)

    ravel = flatten

    def copy(self):
        return Sym(self._shape, self.desc)

    # ---- anything numeric aborts the trace
    def _abort(self, *a, **k):
        raise TraceAbort("arithmetic on the symbolic probe")

    __array__ = _abort
    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = _abort
    __truediv__ = __rtruediv__ = __neg__ = __pow__ = __abs__ = _abort
    __iadd__ = __isub__ = __imul__ = __itruediv__ = _abort
    __getitem__ = __setitem__ = __len__ = __iter__ = _abort
    __float__ = __bool__ = _abort


def trace_operator(fn, n):
    """Descriptor of fn applied to a flat length-n probe, or None.
    ("identity",) if fn returns its (reshaped) argument."""
    try:
        out = fn(Sym((n,)))
    except Exception:
        return None
    if not isinstance(out, Sym):
        return None
    if out.desc is None:
        return ("identity",) if out.size == n else None
    return out.desc


def trace_prox(fn, n):
    """Descriptor of a prox callable fn(x, step) or None."""
    tau = TauSym(0.3141592653589793)
    try:
        out = fn(Sym((n,)), tau)
    except Exception:
        return None
    if not isinstance(out, Sym) or out.desc is None:
        return None
    return out.desc
