"""What the trailing one / two iterations of a run cost at 512^3 (float32 TV-L2):
k_pd_fusedk (3 per launch), k_pd_fused2 or k_pd_fusedk at depth 2, k_pd_fused."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from nsol_amd import ops, _lib
from nsol_amd.primal_dual_solver import step_schedule
from nsol_amd.synthetic import synth_volume

n = 512
shape = (n, n, n)
nvox = n ** 3
vol = synth_volume(n, seed=0, kind="gauss", dtype=np.float32)
bt = torch.from_numpy(vol.reshape(-1)).cuda()
bt = ops.scale(bt, float(vol.max()), divide=True)
x, x_alt = bt.clone(), torch.empty_like(bt)
xbar = [bt.clone(), torch.empty_like(bt)]
p = [torch.zeros(3 * nvox, device="cuda") for _ in range(2)]
w = (1.0, 1.0, 1.0)
lm = 1 / 0.03
flags = ops.PD_REG_TV | ops.PD_DATA_L2
sig, ta, th = step_schedule("ALG2", 16.0, lm, 64)
for _ in range(8):
    ops.pd_run(xbar[0], xbar[1], x, bt, p[0], p[1], shape, w, lm, sig[:60], ta[:60], th[:60], True, 0.05, flags, x_alt=x_alt)
    torch.cuda.synchronize()
    if ops.pd_fusedk_tuned(x, shape) != 0:
        break

def t(count, reps=20):
    def f():
        ops.pd_run(xbar[0], xbar[1], x, bt, p[0], p[1], shape, w, lm, sig[:count], ta[:count], th[:count], False, 0.05, flags, x_alt=x_alt, swap_ok=True)
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

for pd2 in (1, 0, 1, 0):
    _lib.set_param("pd2_enable", pd2)
    print("pd2_enable=%d  1 it %.4f ms   2 it %.4f ms   3 it %.4f ms   4 it %.4f   5 it %.4f" % (pd2, t(1), t(2), t(3), t(4), t(5)), flush=True)
