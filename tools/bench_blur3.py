#!/usr/bin/env python3
"""One-pass blur (nsol_corr3_wrap_*) at n^3: the LDS-DMA staged kernel against
the register-window kernel, interleaved, several z-chunk lengths.
    python tools/bench_blur3.py [n | nz,ny,nx] [cov] [float32|float64]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import ops, _lib
import nsol_amd.kernels as K

dims = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512").split(",")]
shape = tuple(dims * 3 if len(dims) == 1 else dims)
nvox = int(np.prod(shape))
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
dt = torch.float64 if len(sys.argv) > 3 and sys.argv[3] == "float64" else torch.float32
taps = K.Kernels1D().get_gaussian(cov)
x = torch.rand(nvox, device="cuda", dtype=dt)
out = torch.empty_like(x)
esize = x.element_size()
# (dma, zchunk, waves per workgroup, stagger)
cfgs = [(1, 0, 16, 0), (0, 0, 16, 0), (1, 64, 16, 0), (1, 171, 16, 0)]
times = {c: [] for c in cfgs}
res = {}
for rnd in range(5):
    for dma, zc, nw, sg in cfgs:
        _lib.set_param("corr_blur3_dma", dma)
        _lib.set_param("corr_blur3_dma_rag", dma)
        _lib.set_param("corr_blur3_zchunk", zc)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            if ops.corr3_wrap(x, shape, taps, taps, taps, out=out) is None:
                raise SystemExit("one-pass kernel does not apply")
        e1.record(); torch.cuda.synchronize()
        if rnd > 0:
            times[(dma, zc, nw, sg)].append(e0.elapsed_time(e1) / 10)
        if (dma, zc, nw, sg) not in res:
            res[(dma, zc, nw, sg)] = out.clone()
_lib.reset_params()
for (dma, zc, nw, sg), ts in times.items():
    ms = float(np.median(ts))
    print(json.dumps({"shape": list(shape), "kernel": "k_blur3_dma" if dma else "k_blur3_wrap_pp",
                      "taps": len(taps), "zchunk": zc, "waves": nw, "stagger": sg, "ms": round(ms, 4),
                      "min_ms": round(float(np.min(ts)), 4),
                      "GBps_algorithmic": round(2.0 * esize * nvox / ms / 1e6, 1),
                      "frac_of_8TBps": round(2.0 * esize * nvox / ms / 1e6 / 8000, 3)}),
          flush=True)
ref = res[(0, 0, 16, 0)]
for c, r in res.items():
    print(json.dumps({"cfg": c, "max_abs_vs_register_kernel":
                      float((r - ref).abs().max())}))
o3 = x
for axis in (0, 1, 2):
    o3 = ops.corr_axis(o3, shape, axis, taps, len(taps) // 2, "wrap")
print(json.dumps({"max_abs_dma_vs_three_passes":
                  float((o3 - res[(1, 0, 16, 0)]).abs().max())}))
