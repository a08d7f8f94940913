"""Time of nsol_tk1_grad_norm_* and of the blur's forms at 512^3 (float32)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from nsol_amd import ops, kernels

n = 512
shape = (n, n, n)
x = torch.randn(n ** 3, device="cuda")
slot = torch.zeros(2, dtype=torch.float64, device="cuda")
w = (1.0, 1.0, 1.0)

def t(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

print("tk1_grad_norm  %.4f ms" % t(lambda: ops.tk1_grad_norm(x, shape, w, result=slot[0:1])))
taps = kernels.Kernels1D().get_gaussian(4.0)
out = torch.empty_like(x)
print("blur           %.4f ms" % t(lambda: ops.corr3_wrap(x, shape, taps, taps, taps, out=out)))
print("blur epilogue  %.4f ms" % t(lambda: ops.corr3_wrap_axpby(x, out, shape, taps, taps, taps, 1.0, 0.0, result=slot[0:1])))
print("blur norms     %.4f ms" % t(lambda: ops.corr3_wrap_norms(x, out, shape, taps, taps, taps, w, slot)))
