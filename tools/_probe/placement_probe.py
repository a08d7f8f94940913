"""Does the relative placement of a kernel's streams in HBM matter?  k_lsmr_hx (four
arrays, seven streams) with the arrays carved from one slab at different strides."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nsol_amd import ops
MiB = 1 << 20
for n1 in (512, 511):
    n = n1 ** 3
    nbytes = n * 4
    slab = torch.empty(4 * (1 << 29) + 64 * MiB, dtype=torch.uint8, device="cuda")
    base = (-slab.data_ptr()) % (2 * MiB)
    def carve(stride):
        out = []
        for k in range(4):
            o = base + k * stride
            out.append(slab[o:o + nbytes].view(torch.float32))
        return out
    torch_like = (nbytes + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
    for label, stride in (("2 MiB rounding (torch)", torch_like), ("512 MiB", 512 * MiB),
                          ("torch + 256 KiB", torch_like + 256 * 1024),
                          ("torch + 1 MiB", torch_like + MiB),
                          ("torch + 4 KiB", torch_like + 4096),
                          ("torch + 64 KiB", torch_like + 65536),
                          ("516 MiB", 516 * MiB), ("520 MiB", 520 * MiB)):
        if 3 * stride + nbytes + base > slab.numel():
            continue
        hbar, x, h, v = carve(stride)
        for t in (hbar, x, h, v):
            t.fill_(0.5)
        ts = []
        for r in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.lsmr_hx_update(hbar, x, h, v, -0.3, 0.2, -0.4, 0.5, sync=False)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        ms = float(np.median(ts[1:]))
        print(json.dumps({"n": n1, "stride": label, "stride_MiB": round(stride / MiB, 3), "ms": round(ms, 4),
                          "GBps": round(28.0 * n / ms / 1e6)}), flush=True)
    del slab
    torch.cuda.empty_cache()
