#!/usr/bin/env python3
"""primal-dual deconvolution (prox_f = prox_linear_least_squares,
interface :257-280; what `nsol_run_deconvolution --solver PD` runs): synthetic
512^3 volume blurred with sigma = 2, TV regulariser, 10 PD iterations with a
10-iteration LSMR solve of the data term each.  Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--iter-max", type=int, default=10)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--set", action="append", default=[], dest="py_set",
                    help="module.NAME=int under nsol_amd (e.g. lsmr.USE_BLUR_NORMS=0)")
    args = ap.parse_args()
    import importlib
    for item in args.py_set:
        path, val = item.split("=")
        mod, name = path.rsplit(".", 1)
        setattr(importlib.import_module("nsol_amd." + mod), name, int(val))
    import torch
    import nsol_amd.linear_operators as LO
    import nsol_amd.primal_dual_solver as pd
    from nsol_amd.proximal_operators import ProximalOperators as prox
    from nsol_amd.synthetic import synth_volume
    n = args.size
    shape = (n, n, n)
    lo = LO.LinearOperators3D()
    A, A_adj = lo.get_gaussian_blurring_operators(np.diag([4.0, 4.0, 4.0]))
    grad, grad_adj = lo.get_gradient_operators()
    Z = (3 * n, n, n)
    clean = synth_volume(n, 0, "clean", dtype=np.float32)
    dev = torch.device("cuda")
    y = A(torch.from_numpy(clean).to(dev))
    y = y + 0.02 * float(y.max()) * torch.randn(y.shape, device=dev,
                                                generator=torch.Generator(device=dev).manual_seed(1))
    y = y.reshape(-1).contiguous()
    xs = float(y.max())
    A_ = lambda x: A(x.reshape(*shape)).flatten()
    Aa_ = lambda x: A_adj(x.reshape(*shape)).flatten()
    D_ = lambda x: grad(x.reshape(*shape)).flatten()
    Da_ = lambda x: grad_adj(x.reshape(*Z)).flatten()
    pf = lambda x, tau: prox.prox_linear_least_squares(
        x=x, tau=tau, A=A_, A_adj=Aa_, b=y, x0=y, iter_max=args.iter_max, x_scale=xs)
    runs = []
    for _ in range(args.repeat):
        s = pd.PrimalDualSolver(prox_f=pf, prox_g_conj=prox.prox_tv_conj, B=D_,
                                B_conj=Da_, L2=16, alpha=0.01, x0=y,
                                iterations=args.iterations, x_scale=xs,
                                dtype=np.float32)
        torch.cuda.synchronize()
        t0 = time.time()
        s.run()
        torch.cuda.synchronize()
        runs.append(time.time() - t0)
    x = s.get_x_device()
    best = min(runs)
    # durations of the C-ABI entries inside one more run (event pairs: nsol_amd/_timing.py)
    from nsol_amd import _timing
    s2 = pd.PrimalDualSolver(prox_f=pf, prox_g_conj=prox.prox_tv_conj, B=D_, B_conj=Da_, L2=16,
                             alpha=0.01, x0=y, iterations=args.iterations, x_scale=xs,
                             dtype=np.float32)
    torch.cuda.synchronize()
    t0 = time.time()
    with _timing.KernelTimer({"lincomb_clip": 2}) as kt:
        s2.run()
        torch.cuda.synchronize()
        wall = time.time() - t0
    table = {k: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 4),
                 "ms_per_run": round(v["total_ms"], 3)}
             for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]["total_ms"])}
    kernel_ms = sum(v["ms_per_run"] for v in table.values())
    print(json.dumps({
        "metric": "PD-deconvolution iterations/sec on %d^3 fp32" % n,
        "value": args.iterations / best, "unit": "PD iterations/s",
        "seconds_per_run": best, "runs": runs, "execution": s.get_execution(),
        "config": {"iterations": args.iterations, "iter_max": args.iter_max,
                   "workload": "synth_volume(%d,0,'clean') blurred sigma=2 + 2%% noise, TV, alpha=0.01" % n},
        "finite": bool(torch.isfinite(x).all()),
        "in_run": {"wall_s": wall, "kernel_ms": kernel_ms, "entries": table}}))


if __name__ == "__main__":
    main()
