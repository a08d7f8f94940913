"""Stencil kernels of config 4 at 512^3 (float32), rows dealt to the XCDs in slabs or not."""
import sys
import torch
sys.path.insert(0, ".")
from nsol_amd import ops, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n, n, n)
N = n ** 3
r = lambda m: torch.rand(m, device="cuda")
x, g, z, out = r(N), r(N), r(N), r(N)
p3, q3 = r(3 * N), r(3 * N)
slot = torch.zeros(1, dtype=torch.float64, device="cuda")
w = (1.0, 1.0, 1.0)

def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

cases = [
    ("tk1_lanczos", lambda: ops.tk1_lanczos(x, g, z, shape, w, 0.1, 0.5, -0.3, -0.2, out=out, result=slot)),
    ("tk1_grad_norm", lambda: ops.tk1_grad_norm(x, shape, w, result=slot)),
    ("lsmr_v_update", lambda: ops.lsmr_v_update(g, p3, x, ops.B_GRAD, shape, w, 0.5, 0.1, -0.5, sync=False)),
    ("lsmr_u_update", lambda: ops.lsmr_u_update(g, x, z, p3, ops.B_GRAD, shape, w, 0.5, 0.1, -0.5, sync=False)),
    ("admm_vw", lambda: ops.admm_vw_update(x, None, p3, None, q3, shape, w, 0.1, 1.0)),
    ("grad", lambda: ops.grad(x, shape, w, out=q3)),
    ("grad_adj", lambda: ops.grad_adj(p3, shape, w, out=out)),
]
for rep in range(2):
    for blocks in (65536, 16384, 4096):
        _lib.set_param("stencil_blocks", blocks)
        print("blocks=%d " % blocks + "  ".join("%s %.4f" % (k, t(f)) for k, f in cases), flush=True)
_lib.set_param("stencil_blocks", 65536)
