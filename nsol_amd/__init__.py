"""nsol_amd -- MI355X-native primal-dual / ADMM hot path with NSoL's callable API.

Module and class names mirror gift-surg/NSoL (nsol.linear_operators,
nsol.proximal_operators, nsol.primal_dual_solver, nsol.admm_linear_solver,
nsol.tikhonov_linear_solver, ...), so caller code switches by changing the
import.  All arithmetic runs in hand-written HIP kernels (libnsol_hip.so, C ABI
in include/nsol_hip.h); there is no CPU fallback.
"""
from .device import set_default_dtype, get_default_dtype  # noqa: F401
from ._caches import invalidate_caches  # noqa: F401

__version__ = "0.1.0"
