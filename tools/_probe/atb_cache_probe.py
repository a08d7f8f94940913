#!/usr/bin/env python3
"""Does A^T b stay cached across the solves of a primal-dual deconvolution?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import nsol_amd.linear_operators as LO
import nsol_amd.tikhonov_linear_solver as tk
from nsol_amd.proximal_operators import ProximalOperators as prox
n = 96
shape = (n, n, n)
lo = LO.LinearOperators3D()
A, A_adj = lo.get_gaussian_blurring_operators(np.diag([4.0, 4.0, 4.0]))
b = torch.rand(n ** 3, device="cuda") + 1.0
xs = float(b.max())
A_ = lambda v: A(v.reshape(*shape)).flatten()
Aa_ = lambda v: A_adj(v.reshape(*shape)).flatten()
orig = tk._adjoint_of_data
def spy(key_op, A_adj_, bb):
    key = tk._atb_cache._key((bb,), id(key_op))
    hit = tk._atb_cache.lookup((bb,), id(key_op))
    print("lookup", key, "hit" if hit is not None else "miss",
          [e[0] for e in tk._atb_cache.entries], flush=True)
    return orig(key_op, A_adj_, bb)
tk._adjoint_of_data = spy
for i in range(3):
    x = b + 0.1 * torch.randn_like(b)
    prox.prox_linear_least_squares(x, 0.5, A_, Aa_, b, b, iter_max=3, x_scale=xs)
# who calls the plain blur in a solve whose A^T b is cached?
import traceback
from nsol_amd import ops
real = ops.corr3_wrap
def spy2(*a, **k):
    print("corr3_wrap called from:", flush=True)
    print("".join(traceback.format_stack(limit=9)[:-1]), flush=True)
    return real(*a, **k)
ops.corr3_wrap = spy2
x = b + 0.1 * torch.randn_like(b)
prox.prox_linear_least_squares(x, 0.5, A_, Aa_, b, b, iter_max=3, x_scale=xs)
