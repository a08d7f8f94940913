#!/usr/bin/env python3
"""Where does temporal blocking start to pay?  pd_run for cubes of several sizes
with (a) the one-iteration kernel only, (b) + k_pd_fused2, (c) + k_pd_fusedk."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import ops, _lib

iters = 60
for n in [int(a) for a in (sys.argv[1:] or ["64", "96", "128", "160", "192", "256", "320"])]:
    shape = (n, n, n)
    nv = n ** 3
    bt = torch.rand(nv, device="cuda")
    x = bt.clone(); xalt = torch.empty_like(bt)
    xb = [bt.clone(), torch.empty_like(bt)]
    p = [torch.zeros(3 * nv, device="cuda") for _ in range(2)]
    sig = np.full(iters, 0.25); ta = np.full(iters, 0.25); th = np.full(iters, 0.9)
    res = {"n": n}
    for name, e2, ek in (("k1", 0, 0), ("k2", 1, 0), ("k3", 1, 1)):
        _lib.set_param("pd2_enable", e2)
        _lib.set_param("pdk_enable", ek)
        ts = []
        for r in range(8):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.pd_run(xb[0], xb[1], x, bt, p[0], p[1], shape, (1., 1., 1.), 33.0,
                       sig, ta, th, False, 0.05, 0, x_alt=xalt)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters * 1e3)
        res[name + "_us_per_iter"] = round(float(np.min(ts[3:])), 2)
    _lib.set_param("pd2_enable", 1); _lib.set_param("pdk_enable", 1)
    print(json.dumps(res), flush=True)
