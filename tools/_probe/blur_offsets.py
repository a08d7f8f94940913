import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nsol_amd import ops, _lib
import nsol_amd.kernels as K
taps = K.Kernels1D().get_gaussian(4.0)
shape = (512, 512, 512)
n = 512 ** 3
pad = 8 << 20   # floats
buf = torch.rand(2 * n + pad + 64, device="cuda")
base_mis = (-buf.data_ptr() // 4) % (1 << 19)   # floats up to the next 2-MiB boundary
x = buf[base_mis:base_mis + n]
def run(x, out, copy=False):
    ts = []
    for r in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            if copy: out.copy_(x)
            else: assert ops.corr3_wrap(x, shape, taps, taps, taps, out=out) is not None
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    return round(float(np.median(ts[1:])), 4)
for kb in [0, 4, 16, 64, 128, 256, 512, 768, 1024, 1536, 2048, 2560, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768 - 4096]:
    off = base_mis + n + kb * 256
    out = buf[off:off + n]
    print(json.dumps({"out_minus_x_KiB_beyond_512MiB": kb, "blur_ms": run(x, out), "copy_ms": run(x, out, True)}), flush=True)
