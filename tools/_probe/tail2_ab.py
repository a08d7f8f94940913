"""A run's trailing pair at 512^3: k_pd_fused2 against depth 2 of k_pd_fusedk seeded from
the settled depth-3 plan (pdk_tail2), interleaved; and a 20-step repetition as the driver
times it."""
import sys, json
import numpy as np, torch
sys.path.insert(0, ".")
from nsol_amd import ops, _lib
from nsol_amd.primal_dual_solver import step_schedule
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n, n, n); nvox = n ** 3
bt = torch.rand(nvox, device="cuda")
x, x_alt = bt.clone(), torch.empty_like(bt)
xbar = [bt.clone(), torch.empty_like(bt)]
p = [torch.zeros(3 * nvox, device="cuda") for _ in range(2)]
w = (1.0, 1.0, 1.0); lm = 1 / 0.03
flags = ops.PD_REG_TV | ops.PD_DATA_L2
sig, ta, th = step_schedule("ALG2", 16.0, lm, 64)
for _ in range(12):
    ops.pd_run(xbar[0], xbar[1], x, bt, p[0], p[1], shape, w, lm, sig[:60], ta[:60], th[:60], True, 0.05, flags, x_alt=x_alt)
    torch.cuda.synchronize()
    if ops.pd_fusedk_tuned(x, shape) != 0:
        break
def t(count, reps=20):
    def f():
        ops.pd_run(xbar[0], xbar[1], x, bt, p[0], p[1], shape, w, lm, sig[:count], ta[:count], th[:count], False, 0.05, flags, x_alt=x_alt, swap_ok=True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for rnd in range(3):
    for tail2 in (1, 0):
        _lib.set_param("pdk_tail2", tail2)
        print(json.dumps({"pdk_tail2": tail2, "pair_ms": round(t(2), 4), "five_ms": round(t(5), 4),
                          "twenty_ms": round(t(20, 10), 4), "depth2_launches": ops.pd_fusedk_launches(2)}), flush=True)
