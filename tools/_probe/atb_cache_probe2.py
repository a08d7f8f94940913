#!/usr/bin/env python3
"""A^T b across the solves of a primal-dual deconvolution set up like tools/bench_pd_deconv.py."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import nsol_amd.linear_operators as LO
import nsol_amd.primal_dual_solver as pd
import nsol_amd.tikhonov_linear_solver as tk
from nsol_amd.proximal_operators import ProximalOperators as prox
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
IM = int(sys.argv[2]) if len(sys.argv) > 2 else 3
shape = (n, n, n)
lo = LO.LinearOperators3D()
A, A_adj = lo.get_gaussian_blurring_operators(np.diag([4.0, 4.0, 4.0]))
grad, grad_adj = lo.get_gradient_operators()
from nsol_amd.synthetic import synth_volume
clean = synth_volume(n, 0, "clean", dtype=np.float32)
y = A(torch.from_numpy(clean).cuda())
y = (y + 0.02 * float(y.max()) * torch.randn(y.shape, device="cuda")).reshape(-1).contiguous()
xs = float(y.max())
A_ = lambda v: A(v.reshape(*shape)).flatten()
Aa_ = lambda v: A_adj(v.reshape(*shape)).flatten()
D_ = lambda v: grad(v.reshape(*shape)).flatten()
Da_ = lambda v: grad_adj(v.reshape(3 * n, n, n)).flatten()
orig = tk._adjoint_of_data
def spy(key_op, A_adj_, bb):
    key = tk._atb_cache._key((bb,), id(key_op))
    hit = tk._atb_cache.lookup((bb,), id(key_op))
    print("lookup", key, "hit" if hit is not None else "miss",
          [e[0] for e in tk._atb_cache.entries], flush=True)
    return orig(key_op, A_adj_, bb)
tk._adjoint_of_data = spy
pf = lambda x, tau: prox.prox_linear_least_squares(x=x, tau=tau, A=A_, A_adj=Aa_, b=y, x0=y,
                                                   iter_max=IM, x_scale=xs)
s = pd.PrimalDualSolver(prox_f=pf, prox_g_conj=prox.prox_tv_conj, B=D_, B_conj=Da_, L2=16,
                        alpha=0.01, x0=y, iterations=4, x_scale=xs, dtype=np.float32)
import traceback
from nsol_amd import ops
real = ops.corr3_wrap
count = [0]
def spy2(*a, **k):
    count[0] += 1
    if count[0] in (2, 3):
        print("corr3_wrap call %d from:" % count[0], flush=True)
        print("".join(traceback.format_stack(limit=8)[:-1]), flush=True)
    return real(*a, **k)
ops.corr3_wrap = spy2
s.run()
