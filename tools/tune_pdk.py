#!/usr/bin/env python3
"""Time the K-iterations-per-pass kernel (k_pd_fusedk) directly through
nsol_pd_fusedk_iter_* for a list of (K, waves, ntx, zchunk) settings, next to
the two-iteration full-row kernel (k_pd_fused2) as the reference line."""
import argparse
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
    ap.add_argument("--nx", type=int, default=0)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--cfg", default="3:16:0:0,3:12:0:0,2:16:0:0",
                    help="comma list of K:waves:ntx:zchunk[:pf2[:split]]")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--w", default="1,1,1", help="inverse spacings wx,wy,wz")
    args = ap.parse_args()
    n = args.size
    shape = (n, n, args.nx or n)
    nv = int(np.prod(shape))
    td = torch.float32 if args.dtype == "f32" else torch.float64
    dev = torch.device("cuda")
    bt = torch.rand(nv, device=dev, dtype=td)
    x = [bt.clone(), torch.empty_like(bt)]
    xb = [bt.clone(), torch.empty_like(bt)]
    p = [torch.zeros(3 * nv, device=dev, dtype=td) for _ in range(2)]
    w = tuple(float(t) for t in args.w.split(","))
    _lib.set_param("pdk_min_kvox", 0)
    cfgs = [tuple(int(t) for t in c.split(":")) for c in args.cfg.split(",")]
    cfgs.append(("pd2", 0, 0, 0))
    times = {c: [] for c in cfgs}
    for rnd in range(args.rounds + 1):
        for c in cfgs:
            k = 2 if c[0] == "pd2" else c[0]
            s = [0.25] * k
            h = [1.0125] * k
            tl = [8.25] * k
            th = [0.9] * k
            if c[0] != "pd2":
                _lib.set_param("pdk_enable", 1)
                _lib.set_param("pdk_nw", c[1])
                _lib.set_param("pdk_ntx", c[2])
                _lib.set_param("pdk_zchunk", c[3])
                _lib.set_param("pdk_pf2", c[4] if len(c) > 4 else -1)
                _lib.set_param("pdk_split", c[5] if len(c) > 5 else -1)
            if c[0] != "pd2":
                _lib.set_param("pdk_verbose", 1 if rnd == 0 else 0)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.launches):
                if i == 1 and c[0] != "pd2":
                    _lib.set_param("pdk_verbose", 0)
                a, b = i & 1, 1 - (i & 1)
                if c[0] == "pd2":
                    ok = ops.pd_fused2_iter(xb[a], xb[b], x[a], x[b], bt, p[a],
                                            p[b], shape, w, s, h, s, tl, th,
                                            args.flags)
                else:
                    ok = ops.pd_fusedk_iter(xb[a], xb[b], x[a], x[b], bt, p[a],
                                            p[b], shape, w, s, h, s, tl, th,
                                            args.flags)
                if not ok:
                    raise SystemExit("kernel does not apply: %r" % (c,))
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[c].append(e0.elapsed_time(e1) / (args.launches * k))
    _lib.set_param("pdk_enable", 0)
    for c in cfgs:
        ms = float(np.median(times[c]))
        print(json.dumps({"cfg": c, "ms_per_iter": round(ms, 4),
                          "min_ms": round(float(np.min(times[c])), 4),
                          "alg_GBps": round(11 * bt.element_size() * nv / ms / 1e6, 1),
                          "it_per_s": round(1e3 / ms, 1)}), flush=True)


if __name__ == "__main__":
    main()
