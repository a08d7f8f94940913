#!/usr/bin/env python3
"""Latency of reading a kernel's scalars back: device tensor + .cpu() against a kernel that
writes into pinned (device-visible) host memory + a stream synchronisation."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nsol_amd import _lib
from nsol_amd.device import stream_ptr
lib = _lib.load()
src = torch.arange(8, dtype=torch.float64, device="cuda")
dev = torch.empty(8, dtype=torch.float64, device="cuda")
pin = torch.empty(8, dtype=torch.float64).pin_memory()
big = torch.rand(1 << 26, device="cuda")
def kern(dst):
    _lib.check(lib.nsol_scale_f64(dst.data_ptr(), src.data_ptr(), 2.0, 0, 8, stream_ptr()), "scale")
st = torch.cuda.current_stream()
def a():
    kern(dev); return dev.cpu().numpy()[3]
def b():
    kern(pin); st.synchronize(); return pin.numpy()[3]
def c():
    kern(dev); return float(dev[3].item())
for name, f in (("device + .cpu()", a), ("pinned + stream sync", b), ("device + .item()", c)):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(2000): v = f()
    dt = (time.perf_counter() - t0) / 2000
    assert v == 6.0
    print("%-24s %.1f us per read-back (idle GPU)" % (name, dt * 1e6), flush=True)
