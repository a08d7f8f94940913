#!/usr/bin/env python3
"""Time the separable Gaussian passes (sigma = 2, 13 taps, periodic) at 512^3."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import ops, _lib
import nsol_amd.kernels as K

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n, n, n)
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
taps = K.Kernels1D().get_gaussian(cov)
R = len(taps) // 2
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
                ops.corr_axis(x, shape, axis, taps, R, "wrap", out=out)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        ms = float(np.median(ts[1:]))
        print(json.dumps({"ra": ra, "xv": xvv, "axis": axis, "ms": round(ms, 4),
                          "GBps": round(8.0 * n ** 3 / ms / 1e6, 1)}), flush=True)

# the one-pass kernel (x, y, z fused)
_lib.set_param("corr_ra", 8)
_lib.set_param("corr_xv", 1)
ref = None
cfgs = ((16, 0), (16, 64), (16, 171), (32, 0), (8, 0))
times = {c: [] for c in cfgs}
for rnd in range(4):                      # interleaved: boxes drift
    for lxb, want in cfgs:  # want = forced z-chunk, 0 = model
        _lib.set_param("corr_blur3_lxb", lxb)
        _lib.set_param("corr_blur3_zchunk", want)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            if ops.corr3_wrap(x, shape, taps, taps, taps, out=out) is None:
                raise SystemExit("one-pass kernel does not apply")
        e1.record(); torch.cuda.synchronize()
        if rnd > 0:
            times[(lxb, want)].append(e0.elapsed_time(e1) / 10)
        if ref is None:
            ref = out.clone()
        assert float((out - ref).abs().max()) == 0.0
for (lxb, want), ts in times.items():
    ms = float(np.median(ts))
    print(json.dumps({"kernel": "corr3_wrap", "taps": len(taps), "lanes_per_row": lxb, "zchunk": want,
                      "ms": round(ms, 4),
                      "min_ms": round(float(np.min(ts)), 4),
                      "GBps_algorithmic": round(8.0 * n ** 3 / ms / 1e6, 1)}), flush=True)
_lib.set_param("corr_blur3_lxb", 16)
_lib.set_param("corr_blur3_zchunk", 0)
# against the three-pass path
_lib.set_param("corr_ra", 8)
o3 = x
for axis in (0, 1, 2):
    o3 = ops.corr_axis(o3, shape, axis, taps, R, "wrap")
print(json.dumps({"max_abs_one_pass_vs_three_passes": float((o3 - ref).abs().max())}))
