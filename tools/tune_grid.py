#!/usr/bin/env python3
"""Streaming kernels at 512^3 against the grid-stride cap (workgroups in flight):
scale (2 streams), lincomb2 (3), clip (2), dot (2 reads), grad (1 -> 3),
grad_adj (3 -> 1), lincomb3 (4)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import ops, _lib

n = 512 ** 3
a, b, c, o = (torch.rand(n, device="cuda") for _ in range(4))
p3 = torch.rand(3 * n, device="cuda")
o3 = torch.empty(3 * n, device="cuda")
shape = (512, 512, 512)
w = (1.0, 1.0, 1.0)
tests = {
    "scale": (lambda: ops.scale(a, 0.5, out=o), 8),
    "lincomb2": (lambda: ops.lincomb2(0.5, a, 0.25, b, out=o), 12),
    "lincomb3": (lambda: ops.lincomb3(0.5, a, 0.25, b, 0.3, c, out=o), 16),
    "clip": (lambda: ops.clip(a, 0.1, 0.9, out=o), 8),
    "dot": (lambda: ops.dot(a, b), 8),
    "scale3n": (lambda: ops.scale(p3, 0.5, out=o3), 24),
}
caps = (4096, 2048, 1024, 768, 512, 256)
times = {(k, cap): [] for k in tests for cap in caps}
for rnd in range(4):
    for cap in caps:
        _lib.set_param("max_grid_blocks", cap)
        for k, (fn, _) in tests.items():
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            if rnd:
                times[(k, cap)].append(e0.elapsed_time(e1) / 10)
_lib.set_param("max_grid_blocks", 2048)
for k, (_, bpv) in tests.items():
    row = {"kernel": k}
    for cap in caps:
        ms = float(np.median(times[(k, cap)]))
        row[str(cap)] = round(bpv * n / ms / 1e6)
    print(json.dumps(row), flush=True)
