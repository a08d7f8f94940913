#!/usr/bin/env python3
"""When do the workgroups of the lean second Lanczos half reach phase 0 and phase N?
(library built with -DNSOL_B3_DRIFT_PROBE=N: wall_clock64 at 100 MHz into the workspace)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nsol_amd import ops
import nsol_amd.kernels as K
n = 512
shape = (n, n, n)
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda: torch.rand(n ** 3, device="cuda", generator=g)
y, yp, t, yn = r(), r(), r(), r()
lb = ops.LanczosBoard(y, 8, 0.1, 0.0)
lb.board[0:1] = ops.dot(y, y); lb.board[3:4] = ops.dot(y, y)
lb.init()
taps = K.Kernels1D().get_gaussian(4.0)
ws, _ = ops._workspace(y.device)
tiles = 256
for rep in range(4):
    ops.corr3_lanczos_b2(t, y, yp, yn, shape, taps, taps, taps, lb, 1)
    torch.cuda.synchronize()
    w = ws[:4 * tiles].cpu().numpy()
    t0, tn = w[3 * tiles:4 * tiles] / 100.0, w[2 * tiles:3 * tiles] / 100.0   # microseconds
    base = t0.min()
    # logical tile l runs on XCD l // 32 (per_xcd consecutive tiles)
    print("rep %d: start spread %.1f us; phase-N spread %.1f us (mean %.1f after first start); "
          "per XCD phase-N spread: %s" % (
              rep, t0.max() - t0.min(), tn.max() - tn.min(), tn.mean() - base,
              " ".join("%.1f" % (tn[k * 32:(k + 1) * 32].max() - tn[k * 32:(k + 1) * 32].min())
                       for k in range(8))), flush=True)
    if rep == 3:
        d = tn - base
        print("phase-N time by LOGICAL tile (us), rows of 8 (dispatch slot fastest), z chunk 0:")
        for ty in range(8):
            print("  " + " ".join("%6.1f" % d[ty * 8 + tx] for tx in range(8)))
