"""Which kernel should run a trailing PAIR of iterations at 512^3: k_pd_fused2 (full
rows) or k_pd_fusedk<K = 2> (tiled footprints)?  And what does the copy-back of x cost?"""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nsol_amd import ops, _lib
from nsol_amd.primal_dual_solver import step_schedule
shape = (512, 512, 512)
n = 512 ** 3
bt = torch.rand(n, device="cuda")
x = bt.clone(); xa = torch.empty_like(bt)
xb = [bt.clone(), torch.empty_like(bt)]
p = [torch.zeros(3 * n, device="cuda") for _ in range(2)]
flags = ops.PD_REG_TV | ops.PD_DATA_L2
for iters in (2, 4, 3, 6, 20):
    sig, ta, th = step_schedule("ALG2", 16.0, 1 / 0.03, iters)
    for pd2 in (1, 0):
        _lib.set_param("pd2_enable", pd2)
        ts = []
        for r in range(12):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.pd_run(xb[0], xb[1], x, bt, p[0], p[1], shape, (1.0, 1.0, 1.0), 1 / 0.03,
                       sig, ta, th, False, 0.05, flags, x_alt=xa)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(json.dumps({"iterations": iters, "pd2_enable": pd2, "ms_per_run": round(float(np.median(ts[4:])), 4),
                          "k2_launches": ops.pd_fusedk_launches(2), "k3_launches": ops.pd_fusedk_launches(3)}), flush=True)
_lib.reset_params()
