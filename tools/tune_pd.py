#!/usr/bin/env python3
"""Sweep the fused primal-dual kernel's tuning knobs on one GPU (interleaved
rounds in ONE process; reports median ms/iteration and algorithmic GB/s)."""
import argparse
import itertools
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import ops, _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--nx", type=int, default=0, help="x extent (default size)")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--ry", default="1,2,4")
    ap.add_argument("--zchunk", default="0,8,16,32,64,128")
    ap.add_argument("--two-pass", action="store_true")
    ap.add_argument("--xcd", default="0,1")
    ap.add_argument("--pd2", default="", help="pd2 zchunk list, e.g. 0,16,32")
    ap.add_argument("--pd2-variant", default="0")
    ap.add_argument("--pd2-xcd", default="1")
    args = ap.parse_args()
    n = args.size
    shape = (n, n, args.nx or n)
    nv = n * n * (args.nx or n)
    dev = torch.device("cuda")
    bt = torch.rand(nv, device=dev)
    x = bt.clone()
    xb = [bt.clone(), torch.empty_like(bt)]
    p = [torch.zeros(3 * nv, device=dev) for _ in range(2)]
    sig = np.full(args.iters, 0.25)
    ta = np.full(args.iters, 0.25)
    th = np.full(args.iters, 0.9)
    variants = [(int(r), int(z), 0, int(m)) for r, z, m in itertools.product(
        args.ry.split(","), args.zchunk.split(","), args.xcd.split(","))]
    if args.two_pass:
        variants.append((2, 0, 1, 0))
    for var in [int(t) for t in args.pd2_variant.split(",")]:
        for z in [int(t) for t in args.pd2.split(",") if t != ""]:
            for m in [int(t) for t in args.pd2_xcd.split(",")]:
                variants.append(("pd2", z, m, var))
    x_alt = torch.empty_like(x)
    times = {v: [] for v in variants}
    for rnd in range(args.rounds + 1):
        for v in variants:
            if v[0] == "pd2":
                _lib.set_param("pd2_enable", 1)
                _lib.set_param("pd2_zchunk", v[1])
                _lib.set_param("pd2_variant", v[3])
                _lib.set_param("pd2_xcd_map", v[2])
            else:
                _lib.set_param("pd2_enable", 0)
                _lib.set_param("pd_ry", v[0])
                _lib.set_param("pd_zchunk", v[1])
            _lib.set_param("pd_two_pass", 0 if v[0] == "pd2" else v[2])
            _lib.set_param("pd_xcd_map", v[3])
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.pd_run(xb[0], xb[1], x, bt, p[0], p[1], shape, (1., 1., 1.),
                       33.0, sig, ta, th, False, 0.05, 0, x_alt=x_alt)
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[v].append(e0.elapsed_time(e1) / args.iters)
    for v in variants:
        ms = float(np.median(times[v]))
        print(json.dumps({"ry": v[0], "zchunk": v[1], "two_pass": v[2], "xcd_map": v[3],
                          "ms_per_iter": round(ms, 4),
                          "min_ms": round(float(np.min(times[v])), 4),
                          "alg_GBps": round(44.0 * nv / ms / 1e6, 1),
                          "it_per_s": round(1e3 / ms, 1)}), flush=True)


if __name__ == "__main__":
    main()
