#!/usr/bin/env python3
"""Persistent kernel (one launch per run) against one launch per iteration on
cache-resident problems: wall time of ops.pd_run + synchronize against the
iteration count, to separate the per-run from the per-iteration cost."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import ops
from nsol_amd.primal_dual_solver import step_schedule

for shape in [(64, 64, 64), (256, 256), (96, 96, 96), (32, 32, 32), (1024, 1024)]:
    n = int(np.prod(shape)); d = len(shape)
    for dt in (torch.float32, torch.float64):
        bt = torch.rand(n, device="cuda", dtype=dt)
        rows = {}
        for iters in (50, 200, 800):
            sig, ta, th = step_schedule("ALG2", 4.0 * d, 1 / 0.03, iters)
            for persist in (True, False):
                ops.PD_PERSIST = False
                best = 1e9
                for _ in range(6):
                    x = bt.clone(); xb = [bt.clone(), torch.empty_like(bt)]
                    p = [torch.zeros(d * n, device="cuda", dtype=dt) for _ in range(2)]
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    if persist:
                        ok = ops.pd_persist_run(xb[0], x, bt, p[0], shape, (1.0, 1.0, 1.0),
                                                1 / 0.03, sig, ta, th, True, 0.05,
                                                ops.PD_REG_TV | ops.PD_DATA_L2)
                    if not persist or not ok:
                        ops.pd_run(xb[0], xb[1], x, bt, p[0], p[1], shape, (1.0, 1.0, 1.0),
                                   1 / 0.03, sig, ta, th, True, 0.05,
                                   ops.PD_REG_TV | ops.PD_DATA_L2)
                    torch.cuda.synchronize()
                    ops.drain_persist_checks()
                    best = min(best, time.perf_counter() - t0)
                rows[(iters, persist)] = best * 1e6
        per_it = {p: (rows[(800, p)] - rows[(200, p)]) / 600 for p in (True, False)}
        fixed = {p: rows[(200, p)] - 200 * per_it[p] for p in (True, False)}
        print(json.dumps({"shape": shape, "dtype": str(dt).split(".")[-1],
                          "persist_us_per_iteration": round(per_it[True], 2),
                          "persist_us_per_run": round(fixed[True], 1),
                          "launches_us_per_iteration": round(per_it[False], 2),
                          "launches_us_per_run": round(fixed[False], 1),
                          "run_us_200_iterations": [round(rows[(200, True)], 1),
                                                    round(rows[(200, False)], 1)]}), flush=True)
ops.PD_PERSIST = True
