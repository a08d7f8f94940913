#!/usr/bin/env python3
"""Launch the one-pass blur a few times (profiling target):
    python3 tools/run_blur3.py [n] [launches] [dma 0|1] [float32|float64]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import ops, _lib
import nsol_amd.kernels as K

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dt = torch.float64 if len(sys.argv) > 4 and sys.argv[4] == "float64" else torch.float32
taps = K.Kernels1D().get_gaussian(4.0)
x = torch.rand(n ** 3, device="cuda", dtype=dt)
out = torch.empty_like(x)
_lib.set_param("corr_blur3_dma", dma)
for _ in range(reps):
    assert ops.corr3_wrap(x, (n, n, n), taps, taps, taps, out=out) is not None
torch.cuda.synchronize()
print("ok", float(out[:1000].sum()))
