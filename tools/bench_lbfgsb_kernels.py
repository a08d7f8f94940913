#!/usr/bin/env python3
"""Launch time and effective bandwidth of the O(n) kernels of the GPU-resident
L-BFGS-B at 512^3 float32, for c = 2, 5, 10 stored pairs (host read-backs
included: they are part of what a call costs the solver)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import _lib
from nsol_amd.lbfgsb_device import DeviceBackend

n = (int(sys.argv[1]) if len(sys.argv) > 1 else 512) ** 3
be = DeviceBackend()
gen = torch.Generator(device="cuda").manual_seed(0)
r = lambda: torch.rand(n, device="cuda", generator=gen)
x, g, z = r(), r() - 0.5, r()
free = (torch.rand(n, device="cuda", generator=gen) < 0.2).to(torch.int8)


def timed(fn, reps=6):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


for c in (2, 5, 8, 10):
    ws = [r() for _ in range(c)]
    wy = [r() for _ in range(c)]
    coef = list(np.linspace(0.1, 1.0, c))
    rows = [
        ("mdots(ws, v)", lambda: be.dots(ws, x), 4.0 * (c + 1)),
        ("mdots(ws, v, free)", lambda: be.dots(ws, x, free), 4.0 * (c + 1) + 1),
        ("masked_grams", lambda: be.masked_grams(ws, wy, free), 8.0 * c + 1),
        ("masked_grams (register-staged kernel)", lambda: (
            _lib.set_param("lb_gram_dma", 0), be.masked_grams(ws, wy, free),
            _lib.set_param("lb_gram_dma", 1)), 8.0 * c + 1),
        ("masked_grams (VALU products, LDS-DMA tiles)", lambda: (
            _lib.set_param("lb_gram_mfma", 0), be.masked_grams(ws, wy, free),
            _lib.set_param("lb_gram_mfma", 1)), 8.0 * c + 1),
        ("masked_grams_rgrad", lambda: be.masked_grams_rgrad(
            ws, wy, free, z, x, g, 0.7, coef, coef), 8.0 * c + 1 + 16.0),
        ("masked_grams_rgrad (VALU products)", lambda: (
            _lib.set_param("lb_gram_mfma", 0), be.masked_grams_rgrad(
                ws, wy, free, z, x, g, 0.7, coef, coef),
            _lib.set_param("lb_gram_mfma", 1)), 8.0 * c + 1 + 16.0),
        ("reduced_gradient (wcomb)", lambda: be.reduced_gradient(
            z, x, g, 0.7, ws, wy, coef, coef, free), 4.0 * (3 + 2 * c + 1) + 1),
        ("subspace_direction (wcomb)", lambda: be.subspace_direction(
            z, ws, wy, coef, coef, 0.7, free), 4.0 * (1 + 2 * c + 1) + 1),
    ]
    for name, fn, bpv in rows:
        ms = timed(fn)
        print(json.dumps({"c": c, "op": name, "ms": round(ms, 3),
                          "GBps": round(bpv * n / ms / 1e6)}), flush=True)
    del ws, wy
iw = be.init_where(x, 0.0, float("inf"))
rows = [
    ("projgr", lambda: be.projgr(x, g, 0.0, float("inf")), 8.0),
    ("diff_dots", lambda: be.diff_dots(z, x, g), 16.0),
    ("cauchy_setup", lambda: be.cauchy_setup(x, g, 0.0, float("inf"), iw), 8.0 + 9.0 + 1),
    ("dot", lambda: be.dot(x, g), 8.0),
    ("count_free", lambda: be.count_free(iw), 1.0),
]
for name, fn, bpv in rows:
    ms = timed(fn)
    print(json.dumps({"op": name, "ms": round(ms, 3), "GBps": round(bpv * n / ms / 1e6)}),
          flush=True)
