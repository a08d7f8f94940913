#!/usr/bin/env python3
"""The C-ABI entries of one primal-dual deconvolution run at 512^3 in the order they are
called (which kernels a solve of the data term really costs)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import nsol_amd.linear_operators as LO
import nsol_amd.primal_dual_solver as pd
from nsol_amd import _timing
from nsol_amd.proximal_operators import ProximalOperators as prox
from nsol_amd.synthetic import synth_volume
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
its = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lo = LO.LinearOperators3D()
A, A_adj = lo.get_gaussian_blurring_operators(np.diag([4.0, 4.0, 4.0]))
grad, grad_adj = lo.get_gradient_operators()
dev = torch.device("cuda")
clean = synth_volume(n, 0, "clean", dtype=np.float32)
y = A(torch.from_numpy(clean).to(dev))
y = y + 0.02 * float(y.max()) * torch.randn(y.shape, device=dev,
                                            generator=torch.Generator(device=dev).manual_seed(1))
b = y.reshape(-1).contiguous()
x0 = b
xs = float(b.max())
def run(iters):
    s = pd.PrimalDualSolver(
        prox_f=lambda x, tau: prox.prox_linear_least_squares(
            x=x, tau=tau, A=lambda v: A(v.reshape(n, n, n)).flatten(),
            A_adj=lambda v: A_adj(v.reshape(n, n, n)).flatten(), b=b, x0=x0, iter_max=10, x_scale=xs),
        prox_g_conj=prox.prox_tv_conj,
        B=lambda v: grad(v.reshape(n, n, n)).flatten(),
        B_conj=lambda v: grad_adj(v.reshape(3 * n, n, n)).flatten(),
        L2=16, x0=x0, alpha=0.01, iterations=iters, x_scale=xs, dtype=np.float32)
    s.run()
    return s
run(2)
torch.cuda.synchronize()
with _timing.KernelTimer() as kt:
    run(its)
torch.cuda.synchronize()
for name, e0, e1 in kt.records:
    print("%-32s %.4f" % (name, e0.elapsed_time(e1)))
