#!/usr/bin/env python3
"""BASELINE config 5 through the public API: a batch of independent 512^3 TV-L1
denoising problems, volume i -> rank i % world_size, one gather at the end.

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 \\
        tools/run_batch_tvl1.py [--size 512] [--volumes 8] [--iterations 500]
        [--backend nccl|gloo]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nsol_amd.linear_operators as LO  # noqa: E402
import nsol_amd.primal_dual_solver as pd  # noqa: E402
from nsol_amd.batch import solve_batch  # noqa: E402
from nsol_amd.proximal_operators import ProximalOperators as prox  # noqa: E402
from nsol_amd.synthetic import synth_volume  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--volumes", type=int, default=8)
    ap.add_argument("--iterations", type=int, default=500)
    ap.add_argument("--backend", default="nccl")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(
        torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    n = args.size
    X, Z = (n, n, n), (3 * n, n, n)
    grad, grad_adj = LO.LinearOperators3D().get_gradient_operators()
    D = lambda x: grad(x.reshape(*X)).flatten()
    Da = lambda x: grad_adj(x.reshape(*Z)).flatten()

    # the synthetic inputs are made (on the host) and uploaded before the clock
    # starts: generating one 512^3 salt-and-pepper volume takes longer than
    # solving it
    from nsol_amd.batch import shard_indices
    rank0 = dist.get_rank() if world > 1 else 0
    vols = {i: torch.from_numpy(synth_volume(n, i, "sp", np.float32)
                                .reshape(-1)).cuda()
            for i in shard_indices(args.volumes, rank0, world)}

    def solve_one(i):
        vol = vols[i]
        xs = float(vol.max())
        pf = lambda x, tau: prox.prox_ell1_denoising(x, tau, x0=vol, x_scale=xs)
        s = pd.PrimalDualSolver(prox_f=pf, prox_g_conj=prox.prox_tv_conj, B=D,
                                B_conj=Da, L2=16, x0=vol, alpha=0.6,
                                iterations=args.iterations, x_scale=xs)
        s.run()
        assert s.get_execution() == "fused"
        return s.get_x_device()

    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    out = solve_batch(solve_one, args.volumes)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rank = dist.get_rank() if world > 1 else 0
    if rank == 0:
        ok = all(bool(torch.isfinite(o).all().item()) for o in out)
        print(json.dumps({"volumes": args.volumes, "ranks": world, "size": n,
                          "iterations": args.iterations, "seconds": dt,
                          "volume_iterations_per_s":
                              args.volumes * args.iterations / dt,
                          "gathered": len(out), "finite": ok}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
