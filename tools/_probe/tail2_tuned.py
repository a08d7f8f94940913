"""What depth 2 of k_pd_fusedk settles on at 512^3 when it is allowed to explore (pairs only)."""
import sys, json
import numpy as np, torch
sys.path.insert(0, ".")
from nsol_amd import ops, _lib
from nsol_amd.primal_dual_solver import step_schedule
n = 512; shape = (n, n, n); nvox = n ** 3
bt = torch.rand(nvox, device="cuda")
x, x_alt = bt.clone(), torch.empty_like(bt)
xbar = [bt.clone(), torch.empty_like(bt)]
p = [torch.zeros(3 * nvox, device="cuda") for _ in range(2)]
w = (1.0, 1.0, 1.0); lm = 1 / 0.03
flags = ops.PD_REG_TV | ops.PD_DATA_L2
sig, ta, th = step_schedule("ALG2", 16.0, lm, 64)
_lib.set_param("pdk_tail2", 0); _lib.set_param("pd2_enable", 0); _lib.set_param("pdk_verbose", 2)
def f(count):
    ops.pd_run(xbar[0], xbar[1], x, bt, p[0], p[1], shape, w, lm, sig[:count], ta[:count], th[:count], False, 0.05, flags, x_alt=x_alt, swap_ok=True)
for i in range(400):
    f(2)
    if i % 20 == 0:
        torch.cuda.synchronize()
        if ops.pd_fusedk_tuned(x, shape, 2) == 1: break
torch.cuda.synchronize()
print(json.dumps({"plan_k2": ops.pd_fusedk_plan(x, shape, 2), "launches": i + 1}))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): f(2)
b.record(); torch.cuda.synchronize()
print(json.dumps({"pair_ms_tuned": a.elapsed_time(b) / 20}))
