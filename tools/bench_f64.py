#!/usr/bin/env python3
"""The float64 forms (the reference's own arithmetic type) of the hot kernels at
512^3: primal-dual iterations (TV-L2, TV-L1, Huber-L2), the one-pass blur, ADMM +
LSMR -- next to float32.  One JSON line each."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import ops
from nsol_amd.primal_dual_solver import step_schedule
import nsol_amd.kernels as K

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n1, n1, n1)
n = n1 ** 3
for td, name in ((torch.float32, "f32"), (torch.float64, "f64")):
    bt = torch.rand(n, device="cuda", dtype=td)
    x = bt.clone(); xa = torch.empty_like(bt)
    xb = [bt.clone(), torch.empty_like(bt)]
    p = [torch.zeros(3 * n, device="cuda", dtype=td) for _ in range(2)]
    iters = 120
    sig, ta, th = step_schedule("ALG2", 16.0, 1 / 0.03, iters)
    for label, flags in (("TV-L2", ops.PD_REG_TV | ops.PD_DATA_L2),
                         ("TV-L1", ops.PD_REG_TV | ops.PD_DATA_L1),
                         ("Huber-L2", ops.PD_REG_HUBER | ops.PD_DATA_L2)):
        ts = []
        for r in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.pd_run(xb[0], xb[1], x, bt, p[0], p[1], shape, (1.0, 1.0, 1.0), 1 / 0.03,
                       sig, ta, th, True, 0.05, flags, x_alt=xa)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters)
        ms = float(np.median(ts[2:]))
        esz = bt.element_size()
        print(json.dumps({"what": "pd " + label, "dtype": name, "ms_per_iteration": round(ms, 4),
                          "iterations_per_s": round(1e3 / ms, 1),
                          "GBps_algorithmic": round(11 * esz * n / ms / 1e6, 1)}), flush=True)
    del x, xa, xb, p
    taps = K.Kernels1D().get_gaussian(4.0)
    out = torch.empty_like(bt)
    ts = []
    for r in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.corr3_wrap(bt, shape, taps, taps, taps, out=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = float(np.median(ts[1:]))
    print(json.dumps({"what": "blur 13 taps", "dtype": name, "ms": round(ms, 4),
                      "GBps_algorithmic": round(2 * bt.element_size() * n / ms / 1e6, 1)}), flush=True)
    del bt, out
    torch.cuda.empty_cache()
