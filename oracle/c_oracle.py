"""ctypes loader of the C restatement oracle/pd_oracle.c.

TEST INFRASTRUCTURE ONLY (same rule as oracle/nsol_oracle.py): imported by
tests/, __graft_entry__ and bench.py's cpu_baseline leg, never by nsol_amd.
`build()` compiles the file with gcc into oracle/_build/ (git-ignored; it
travels to the GPU box like the HIP library)."""
import ctypes
import os
import shutil
import subprocess

import numpy as np

from . import nsol_oracle as _np_oracle

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "pd_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libpd_oracle.so")
_lib = None


def build(force=False):
    if not force and os.path.exists(LIB) and \
            os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found: cannot build oracle/pd_oracle.c")
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = LIB + ".tmp%d" % os.getpid()
    subprocess.run([gcc, "-O2", "-fopenmp", "-ffp-contract=off", "-fPIC",
                    "-shared", SRC, "-o", tmp, "-lm"], check=True)
    os.replace(tmp, LIB)
    return LIB


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        P = ctypes.c_void_p
        lib.orc_pd_denoise.restype = ctypes.c_int
        lib.orc_pd_denoise.argtypes = [
            P, P, P, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, P, ctypes.c_double, ctypes.c_int, ctypes.c_int,
            ctypes.c_double, P, P, P, ctypes.c_int]
        _lib = lib
    return _lib


def threads():
    """OpenMP threads the library will use (OMP_NUM_THREADS or all cores)."""
    v = os.environ.get("OMP_NUM_THREADS")
    return int(v) if v else (os.cpu_count() or 1)


def primal_dual_denoise(b, shape, reg="TV", data="L2", alpha=0.03,
                        iterations=10, L2=8., alg_type="ALG2", x_scale=None,
                        spacing=None, x0=None):
    """Same signature and result as nsol_oracle.primal_dual_denoise."""
    lib = _load()
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
    x_scale = float(np.max(b)) if x_scale is None else float(x_scale)
    x0 = b if x0 is None else \
        np.ascontiguousarray(x0, dtype=np.float64).reshape(-1)
    d = len(shape)
    h = np.ones(d) if spacing is None else \
        np.ascontiguousarray(spacing, dtype=np.float64)
    lmbda = 1. / float(alpha)
    sig, ta, th = _np_oracle.pd_schedule(alg_type, L2, lmbda, iterations)
    sig, ta, th = (np.ascontiguousarray(a, dtype=np.float64)
                   for a in (sig, ta, th))
    ext = (1,) * (3 - d) + tuple(int(s) for s in shape)
    out = np.empty_like(b)
    rc = lib.orc_pd_denoise(
        out.ctypes.data, b.ctypes.data, x0.ctypes.data, d, ext[0], ext[1],
        ext[2], h.ctypes.data, x_scale, {"TV": 0, "Huber": 1}[reg],
        {"L2": 0, "L1": 1}[data], lmbda, sig.ctypes.data, ta.ctypes.data,
        th.ctypes.data, int(iterations))
    if rc != 0:
        raise MemoryError("orc_pd_denoise")
    return out
