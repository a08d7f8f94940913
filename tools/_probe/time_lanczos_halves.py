#!/usr/bin/env python3
"""Durations of the blur's forms at 512^3 / 13 taps (float32), interleaved (each launch
behind a different kernel, as in a solve), median of reps."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nsol_amd import ops
import nsol_amd.kernels as K
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
shape = (n, n, n)
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda: torch.rand(n ** 3, device="cuda", generator=g)
y, yp, t, q0, yn = r(), r(), r(), r(), r()
lb = ops.LanczosBoard(y, 8, 0.1, 0.0)
lb.board[0:1] = ops.dot(y, y); lb.board[3:4] = ops.dot(y, y)
lb.init()
taps = K.Kernels1D().get_gaussian(4.0)
fns = {"EPI0 blur": lambda: ops.corr3_wrap(y, shape, taps, taps, taps, out=t),
       "EPI2 a2": lambda: ops.corr3_lanczos_a2(y, t, shape, taps, taps, taps, lb, 1),
       "EPI6 b2": lambda: ops.corr3_lanczos_b2(t, y, yp, yn, shape, taps, taps, taps, lb, 1),
       "EPI6 b2 no y_prev": lambda: ops.corr3_lanczos_b2(t, y, None, yn, shape, taps, taps, taps, lb, 1),
       "EPI3 a": lambda: ops.corr3_lanczos_a(y, yp, t, q0, shape, taps, taps, taps, lb, 1),
       "EPI4 b": lambda: ops.corr3_lanczos_b(t, q0, y, yn, shape, taps, taps, taps, lb, 1)}
for f in fns.values():
    f(); f()
times = {k: [] for k in fns}
for _ in range(reps):
    for k, f in fns.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record()
        times[k].append((e0, e1))
torch.cuda.synchronize()
print(" | ".join("%s %.4f" % (k, sorted(a.elapsed_time(b) for a, b in v)[len(v) // 2])
                 for k, v in times.items()), flush=True)
