import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nsol_amd import ops, _lib
import nsol_amd.kernels as K
taps = K.Kernels1D().get_gaussian(4.0)
for shape in [(512, 512, 512), (512, 512, 511)]:
    n = int(np.prod(shape))
    xb = torch.rand(n + 8, device="cuda"); ob = torch.empty(n + 8, device="cuda")
    for xo, oo in [(0, 0), (1, 0), (0, 1), (1, 1), (2, 2)]:
        x = xb[xo:xo + n]; out = ob[oo:oo + n]
        ts = []
        for r in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                assert ops.corr3_wrap(x, shape, taps, taps, taps, out=out) is not None
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        print(json.dumps({"shape": shape, "x_off": xo, "out_off": oo, "ms": round(float(np.median(ts[1:])), 4)}), flush=True)
