#!/usr/bin/env python3
"""Launch durations of the blur's forms at 512^3 / 13 taps (float32)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nsol_amd import ops
import nsol_amd.linear_operators as LO
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n, n, n)
A, _ = LO.LinearOperators3D().get_gaussian_blurring_operators(np.diag([4.0] * 3))
half_a, half_b = A.lanczos_halves(shape)
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda: torch.rand(n ** 3, device="cuda", generator=g)
y, yp, t, q0, yn, z = r(), r(), r(), r(), r(), torch.zeros(n ** 3, device="cuda")
lb = ops.LanczosBoard(y, 8, 0.1, 0.0)
lb.board[0:1] = ops.dot(y, y); lb.board[3:4] = ops.dot(y, y)
lb.init()
sums = torch.zeros(2, dtype=torch.float64, device="cuda")
slot = torch.zeros(1, dtype=torch.float64, device="cuda")
w = (1., 1., 1.)
fns = {"blur": lambda: A(y.view(shape)),
       "norms (EPI 2)": lambda: A.apply_norms(y, t, shape, w, sums),
       "axpby (EPI 1)": lambda: A.apply_axpby(y, t, shape, 1.0, 0.5, result=slot),
       "half_a (EPI 3)": lambda: half_a(y, yp, t, q0, lb, 1),
       "half_a no prev": lambda: half_a(y, None, t, q0, lb, 1),
       "half_b (EPI 4)": lambda: half_b(t, q0, y, yn, lb, 1),
       "tk1_lanczos": lambda: ops.tk1_lanczos(y, t, yp, shape, w, 0.1, 0.5, -0.3, -0.2, out=yn, result=slot)}
for _ in range(20): fns["blur"]()
for name, fn in fns.items():
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"kernel": name, "ms": round(e0.elapsed_time(e1) / 20, 4)}), flush=True)
