#!/usr/bin/env python3
"""Secondary benchmark (BASELINE.json configs[3]): synthetic 512^3 Gaussian-blur
deconvolution (sigma = 2, 13 taps per axis, periodic) with ADMMLinearSolver,
TK1-regularised inner problem.  Prints one JSON line.

    python bench_admm.py [--size 512] [--iterations 10] [--iter-max 10]
                         [--minimizer lsmr|L-BFGS-B] [--data-loss linear|huber]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--iter-max", type=int, default=10)
    ap.add_argument("--minimizer", default="lsmr")
    ap.add_argument("--data-loss", default="linear")
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()

    import torch
    import nsol_amd.linear_operators as LO
    import nsol_amd.admm_linear_solver as admm
    from nsol_amd import ops
    from nsol_amd.synthetic import synth_volume

    n = args.size
    shape = (n, n, n)
    lo = LO.LinearOperators3D()
    A, A_adj = lo.get_gaussian_blurring_operators(np.diag([4.0, 4.0, 4.0]))
    grad, grad_adj = lo.get_gradient_operators()
    Z = (3 * n, n, n)
    A_1D = lambda x: A(x.reshape(*shape)).flatten()
    A_adj_1D = lambda x: A_adj(x.reshape(*shape)).flatten()
    D_1D = lambda x: grad(x.reshape(*shape)).flatten()
    D_adj_1D = lambda x: grad_adj(x.reshape(*Z)).flatten()

    clean = torch.from_numpy(synth_volume(n, 0, "clean", np.float32)).cuda()
    y = A(clean).flatten()
    gen = torch.Generator(device="cuda").manual_seed(1)
    y = y + 0.02 * float(y.max()) * torch.randn(y.shape, device="cuda",
                                                 generator=gen)
    x_scale = float(y.max())
    times = []
    for _ in range(args.repeat):
        s = admm.ADMMLinearSolver(
            A=A_1D, A_adj=A_adj_1D, b=y, B=D_1D, B_adj=D_adj_1D, x0=y,
            dimension=3, alpha=0.01, rho=0.1, iterations=args.iterations,
            iter_max=args.iter_max, minimizer=args.minimizer,
            data_loss=args.data_loss, x_scale=x_scale, dtype=np.float32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    x = s.get_x_device()
    rel_change = float((ops.norm2(ops.lincomb2(1.0, x, -1.0, y)) /
                        ops.norm2(y)))
    best = min(times)
    print(json.dumps({
        "metric": "ADMM iterations/sec on %d^3 fp32 TV deconvolution" % n,
        "value": args.iterations / best, "unit": "ADMM iterations/s",
        "seconds_per_run": best, "runs": times,
        "config": {"workload": "synth_volume(%d,0,'clean') blurred sigma=2 + "
                               "2%% noise; ADMM alpha=0.01 rho=0.1" % n,
                   "iterations": args.iterations, "iter_max": args.iter_max,
                   "minimizer": args.minimizer, "data_loss": args.data_loss,
                   "execution": s.get_execution()},
        "rel_change_vs_input": rel_change,
        "finite": bool(torch.isfinite(x).all().item())}))


if __name__ == "__main__":
    main()
