#!/usr/bin/env python3
"""The blur's forms one after the other at 512^3 / 13 taps (float32), for rocprofv3 passes:
plain (EPI 0), with sums (EPI 2 = first Lanczos half, lean), second half lean (EPI 6),
second half with q0 (EPI 4), first half with q0 (EPI 3)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nsol_amd import ops
import nsol_amd.linear_operators as LO
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
shape = (n, n, n)
A, _ = LO.LinearOperators3D().get_gaussian_blurring_operators(np.diag([4.0] * 3))
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda: torch.rand(n ** 3, device="cuda", generator=g)
y, yp, t, q0, yn = r(), r(), r(), r(), r()
lb = ops.LanczosBoard(y, 8, 0.1, 0.0)
lb.board[0:1] = ops.dot(y, y); lb.board[3:4] = ops.dot(y, y)
lb.init()
import nsol_amd.kernels as K
taps = K.Kernels1D().get_gaussian(4.0)
for _ in range(reps):
    ops.corr3_wrap(y, shape, taps, taps, taps, out=t)
    ops.corr3_lanczos_a2(y, t, shape, taps, taps, taps, lb, 1)
    ops.corr3_lanczos_b2(t, y, yp, yn, shape, taps, taps, taps, lb, 1)
    ops.corr3_lanczos_a(y, yp, t, q0, shape, taps, taps, taps, lb, 1)
    ops.corr3_lanczos_b(t, q0, y, yn, shape, taps, taps, taps, lb, 1)
torch.cuda.synchronize()
print("done")
