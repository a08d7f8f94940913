#!/usr/bin/env python3
"""k_wcomb (limited-memory combination of L-BFGS-B: 3 + 2c input streams) at 512^3
against the number of workgroups in flight."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import _lib
from nsol_amd.lbfgsb_device import DeviceBackend

n = 512 ** 3
c = int(sys.argv[1]) if len(sys.argv) > 1 else 10
be = DeviceBackend()
z, x, g = (torch.rand(n, device="cuda") for _ in range(3))
ws = [torch.rand(n, device="cuda") for _ in range(c)]
wy = [torch.rand(n, device="cuda") for _ in range(c)]
free = torch.zeros(n, dtype=torch.int8, device="cuda")
coef = list(np.linspace(0.1, 1.0, c))
caps = (4096, 2048, 1024, 768, 512, 256)
times = {cap: [] for cap in caps}
for rnd in range(4):
    for cap in caps:
        _lib.set_param("max_grid_blocks", cap)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            be.reduced_gradient(z, x, g, 0.7, ws, wy, coef, coef, free)
        e1.record(); torch.cuda.synchronize()
        if rnd:
            times[cap].append(e0.elapsed_time(e1) / 5)
_lib.set_param("max_grid_blocks", 2048)
for cap in caps:
    ms = float(np.median(times[cap]))
    print(json.dumps({"stored_pairs": c, "workgroups": cap, "ms": round(ms, 4),
                      "GBps": round((4.0 * (3 + 2 * c + 1) + 1) * n / ms / 1e6)}), flush=True)
