#!/usr/bin/env python3
"""nsol_admm_vw_update_g_* against the two kernels at 512^3 float32 (median of 10)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nsol_amd import ops
n = 512; shape = (n, n, n); N = n ** 3
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda m: torch.randn(m, device="cuda", generator=g)
x, atb, w0, w1, rhs, gv = 3 * r(N), r(N), r(3 * N), torch.empty(3 * N, device="cuda"), torch.empty(3 * N, device="cuda"), torch.empty(N, device="cuda")
sums = torch.zeros(2, dtype=torch.float64, device="cuda")
wgt = (1., 1., 1.)
def med(f, reps=10):
    f(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); ts.append((e0, e1))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ts)[len(ts) // 2]
one = med(lambda: ops.admm_vw_update_g(x, w0, w1, None, atb, gv, shape, wgt, 0.7, 0.3, 1.0, 0.3, sums))
two_a = med(lambda: ops.admm_vw_update(x, None, w0, None, rhs, shape, wgt, 0.7, 0.3))
two_b = med(lambda: ops.lsmr_v_update(atb, rhs, atb, ops.B_GRAD, shape, wgt, 1.0, 0.3, 0.0, out=gv, sync=False))
print("WGS=%s one pass %.4f ms | vw %.4f + v %.4f = %.4f" % (os.environ.get("NSOL_VWG_WGS", "default"), one, two_a, two_b, two_a + two_b), flush=True)
