#!/usr/bin/env python3
"""Time the separable Gaussian passes (sigma = 2, 13 taps, periodic) at 512^3."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import ops, _lib
import nsol_amd.kernels as K

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n, n, n)
taps = K.Kernels1D().get_gaussian(4.0)
x = torch.rand(n ** 3, device="cuda")
out = torch.empty_like(x)
for ra, xvv in ((4, 1), (8, 1), (8, 2)):
    _lib.set_param("corr_ra", ra)
    _lib.set_param("corr_xv", xvv)
    for axis in (0, 1, 2):
        ts = []
        for _ in range(6):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.corr_axis(x, shape, axis, taps, 6, "wrap", out=out)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        ms = float(np.median(ts[1:]))
        print(json.dumps({"ra": ra, "xv": xvv, "axis": axis, "ms": round(ms, 4),
                          "GBps": round(8.0 * n ** 3 / ms / 1e6, 1)}), flush=True)

# the one-pass kernel (x, y, z fused)
_lib.set_param("corr_ra", 8)
_lib.set_param("corr_xv", 1)
for lxb in (32, 16, 64, 8):
    _lib.set_param("corr_blur3_lxb", lxb)
    ts = []
    for _ in range(6):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            assert ops.corr3_wrap(x, shape, taps, taps, taps, out=out) is not None
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    ms = float(np.median(ts[1:]))
    print(json.dumps({"kernel": "corr3_wrap (one pass)", "lanes_per_row": lxb,
                      "ms": round(ms, 4),
                      "GBps_algorithmic": round(8.0 * n ** 3 / ms / 1e6, 1)}), flush=True)
_lib.set_param("corr_blur3_lxb", 16)
