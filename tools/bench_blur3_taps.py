#!/usr/bin/env python3
"""One-pass blur at n^3 for several tap counts and both element types: the LDS-DMA
staged kernel, the register-window kernel and the three separate passes.
    python tools/bench_blur3_taps.py [n]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import ops, _lib
import nsol_amd.kernels as K

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n, n, n)
for dt in (torch.float32, torch.float64):
    x = torch.rand(n ** 3, device="cuda", dtype=dt)
    out = torch.empty_like(x)
    tmp = torch.empty_like(x)
    for cov in (1.0, 4.0, 5.5, 7.0):
        taps = K.Kernels1D().get_gaussian(cov)
        res = {}
        for name in ("dma", "registers", "three_passes"):
            _lib.set_param("corr_blur3_dma", 1 if name == "dma" else 0)
            ts = []
            for rnd in range(4):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    if name == "three_passes":
                        ops.corr_axis(x, shape, 0, taps, len(taps) // 2, "wrap", out=out)
                        ops.corr_axis(out, shape, 1, taps, len(taps) // 2, "wrap", out=tmp)
                        ops.corr_axis(tmp, shape, 2, taps, len(taps) // 2, "wrap", out=out)
                    else:
                        assert ops.corr3_wrap(x, shape, taps, taps, taps, out=out) is not None
                e1.record(); torch.cuda.synchronize()
                if rnd:
                    ts.append(e0.elapsed_time(e1) / 5)
            res[name] = round(float(np.median(ts)), 4)
        _lib.reset_params()
        b = 2.0 * x.element_size() * n ** 3
        print(json.dumps({"dtype": str(dt).split(".")[-1], "taps": len(taps), "ms": res,
                          "dma_frac_of_8TBps": round(b / res["dma"] / 1e6 / 8000, 3)}),
              flush=True)
    del x, out, tmp
