#!/usr/bin/env python3
"""Primal-dual iterations/s (TV-L2, float32, kernels only) for volumes whose
extents are not multiples of the vector width, next to 512^3."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsol_amd import ops
from nsol_amd.primal_dual_solver import step_schedule

shapes = [(512, 512, 512), (511, 511, 511), (512, 512, 510), (512, 510, 512), (513, 513, 516)]
if len(sys.argv) > 1:
    shapes = [tuple(int(t) for t in a.split("x")) for a in sys.argv[1:]]
for shape in shapes:
    n = int(np.prod(shape))
    bt = torch.rand(n, device="cuda")
    x = bt.clone(); xa = torch.empty_like(bt)
    xb = [bt.clone(), torch.empty_like(bt)]
    p = [torch.zeros(3 * n, device="cuda") for _ in range(2)]
    iters = 120
    sig, ta, th = step_schedule("ALG2", 16.0, 1 / 0.03, iters)
    flags = ops.PD_REG_TV | ops.PD_DATA_L2
    # plan pass (as bench.py): the online tuner tries its candidates on the first
    # launches of a new shape; time the settled plan
    for _ in range(12):
        ops.pd_run(xb[0], xb[1], x, bt, p[0], p[1], shape, (1.0, 1.0, 1.0), 1 / 0.03,
                   sig[:60], ta[:60], th[:60], True, 0.05, flags, x_alt=xa)
        torch.cuda.synchronize()
        if ops.pd_fusedk_tuned(x, shape) != 0:
            break
    ts = []
    for r in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.pd_run(xb[0], xb[1], x, bt, p[0], p[1], shape, (1.0, 1.0, 1.0), 1 / 0.03,
                   sig, ta, th, r == 0, 0.05, flags, x_alt=xa)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    ms = float(np.median(ts[1:]))
    print(json.dumps({"shape": shape, "ms_per_iteration": round(ms, 4),
                      "ns_per_voxel": round(ms * 1e6 / n, 4),
                      "plan": ops.pd_fusedk_plan(x, shape),
                      "tuned": ops.pd_fusedk_tuned(x, shape)}), flush=True)
    # the same volume with its rows at a pitch of whole 16-byte vectors (what the solver
    # does for such shapes: nsol_pd_run_pitched_*)
    pitch = ops.row_pitch(shape, bt)
    if pitch:
        btq = ops.to_pitched(bt, shape, pitch)
        del x, xa, xb, p
        xq, xaq = btq.clone(), torch.zeros_like(btq)
        xbq = [btq.clone(), torch.zeros_like(btq)]
        pq = [torch.zeros(3 * btq.numel(), device="cuda") for _ in range(2)]
        for _ in range(12):
            ops.pd_run(xbq[0], xbq[1], xq, btq, pq[0], pq[1], shape, (1.0, 1.0, 1.0), 1 / 0.03,
                       sig[:60], ta[:60], th[:60], True, 0.05, flags, x_alt=xaq, pitch=pitch)
            torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.pd_run(xbq[0], xbq[1], xq, btq, pq[0], pq[1], shape, (1.0, 1.0, 1.0), 1 / 0.03,
                       sig, ta, th, r == 0, 0.05, flags, x_alt=xaq, pitch=pitch)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters)
        print(json.dumps({"shape": shape, "rows_at_pitch": pitch,
                          "ms_per_iteration": round(float(np.median(ts[1:])), 4)}), flush=True)
        del btq, xq, xaq, xbq, pq
        x = xa = xb = p = None
    del bt, x, xa, xb, p
    torch.cuda.empty_cache()
# the one-iteration kernel alone on ragged rows (2-D images and trailing
# iterations): element-aligned 16-byte accesses against the 4-byte form
from nsol_amd import _lib
_lib.set_param("pdk_enable", 0)
_lib.set_param("pd2_enable", 0)
for shape in [(511, 511, 511), (8191, 8190), (512, 512, 512)]:
    n = int(np.prod(shape))
    bt = torch.rand(n, device="cuda")
    x = bt.clone()
    xb = [bt.clone(), torch.empty_like(bt)]
    p = [torch.zeros(len(shape) * n, device="cuda") for _ in range(2)]
    iters = 40
    sig, ta, th = step_schedule("ALG2", 16.0, 1 / 0.03, iters)
    for rag in (1, 0):
        _lib.set_param("pd_rag", rag)
        ts = []
        for r in range(4):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.pd_run(xb[0], xb[1], x, bt, p[0], p[1], shape, (1.0, 1.0, 1.0), 1 / 0.03,
                       sig, ta, th, r == 0, 0.05, ops.PD_REG_TV | ops.PD_DATA_L2)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters)
        ms = float(np.median(ts[1:]))
        print(json.dumps({"kernel": "k_pd_fused", "shape": shape,
                          "ragged_16_byte_accesses": bool(rag),
                          "ms_per_iteration": round(ms, 4),
                          "GBps_algorithmic": round((44 if len(shape) == 3 else 36) * n / ms / 1e6)}),
              flush=True)
    del bt, x, xb, p
    torch.cuda.empty_cache()
_lib.reset_params()
