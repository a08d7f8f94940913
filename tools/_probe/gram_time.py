#!/usr/bin/env python3
"""Launch time of the subspace matrix (with and without the reduced gradient) at 512^3 for
c stored pairs: HIP events around 8 launches (each reads its sums back)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nsol_amd.lbfgsb_device import DeviceBackend  # noqa: E402

n = 512 ** 3
be = DeviceBackend()
gen = torch.Generator(device="cuda").manual_seed(0)
r = lambda: torch.rand(n, device="cuda", generator=gen)
x, g, z = r(), r() - 0.5, r()
free = (torch.rand(n, device="cuda", generator=gen) < 0.2).to(torch.int8)
out = []
for c in (int(a) for a in (sys.argv[1:] or ["10"])):
    ws = [r() for _ in range(c)]
    wy = [r() for _ in range(c)]
    coef = list(np.linspace(0.1, 1.0, c))
    for name, fn in (("gram", lambda: be.masked_grams(ws, wy, free)),
                     ("gram+r", lambda: be.masked_grams_rgrad(ws, wy, free, z, x, g, 0.7, coef, coef))):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(8):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append("c=%d %s %.3f ms" % (c, name, e0.elapsed_time(e1) / 8))
print("; ".join(out), flush=True)
