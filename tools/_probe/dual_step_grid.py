"""k_dual_step / k_primal_step (the two-pass PD forms, PD deconvolution's dual update)
at 512^3 against the element-wise grid cap."""
import sys
import torch
sys.path.insert(0, ".")
from nsol_amd import ops, _lib
n = 512; N = n ** 3; shape = (n, n, n); w = (1.0, 1.0, 1.0)
xbar, x, bt = (torch.rand(N, device="cuda") for _ in range(3))
p, q = torch.rand(3 * N, device="cuda"), torch.empty(3 * N, device="cuda")
xo, xb2 = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")

def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

for rep in range(2):
    for blocks in (1024, 2048, 3072, 4096):
        _lib.set_param("max_grid_blocks", blocks)
        print("max_grid_blocks=%d  dual_step %.4f  grad_adj_axpy %.4f  extrapolate %.4f" % (
            blocks, t(lambda: ops.pd_dual_step(xbar, p, q, shape, w, 0.25, 1.0)),
            t(lambda: ops.grad_adj_axpy(p, x, 0.1, shape, w)),
            t(lambda: ops.extrapolate(x, xbar, 0.5, out=xo))), flush=True)
_lib.set_param("max_grid_blocks", 2048)
