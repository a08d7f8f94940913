#!/usr/bin/env python3
"""BASELINE configs 1 and 2 (cache-resident problems): wall time of
PrimalDualSolver.run() on the GPU and parity against the reference goldens."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import nsol_amd.linear_operators as LO  # noqa: E402
import nsol_amd.primal_dual_solver as pd  # noqa: E402
from nsol_amd.proximal_operators import ProximalOperators as prox  # noqa: E402


def solver(obs, alpha, iters, L2, dtype):
    b = obs.flatten()
    xs = obs.max()
    lo = {2: LO.LinearOperators2D, 3: LO.LinearOperators3D}[obs.ndim]()
    grad, grad_adj = lo.get_gradient_operators()
    X, Z = obs.shape, grad(obs).shape
    D = lambda x: grad(x.reshape(*X)).flatten()
    Da = lambda x: grad_adj(x.reshape(*Z)).flatten()
    pf = lambda x, tau: prox.prox_ell2_denoising(x, tau, x0=b, x_scale=xs)
    return pd.PrimalDualSolver(prox_f=pf, prox_g_conj=prox.prox_tv_conj, B=D,
                               B_conj=Da, L2=L2, x0=b, alpha=alpha,
                               iterations=iters, x_scale=xs, dtype=dtype)


def main():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "configs.npz")))
    lena = g["lena_noise_u8"].astype(np.float64)
    ph = g["phantom64"].astype(np.float64)
    cases = [("config1 Lena 256^2 TVL2 50 it L2=8", lena, 50, 8.0,
              "cfg1_lena_TVL2_50it_L2eq8"),
             ("config2 phantom 64^3 TVL2 200 it L2=16", ph, 200, 16.0,
              "cfg2_phantom_TVL2_200it_L2eq16")]
    from nsol_amd import ops
    for name, obs, iters, L2, key in cases:
        for dtype in (np.float32, np.float64):
            for persist in (True, False):
                ops.PD_PERSIST = persist
                best = 1e9
                before = ops.pd_persist_launches()
                for _ in range(7):
                    s = solver(obs, 0.03, iters, L2, dtype)
                    s._x0_device()                  # upload outside the timing
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    s.run()
                    torch.cuda.synchronize()        # the iterations, not their enqueue
                    best = min(best, time.perf_counter() - t0)
                ref = g[key].astype(np.float64)
                err = np.linalg.norm(s.get_x() - ref) / np.linalg.norm(ref)
                print(json.dumps({"case": name, "dtype": np.dtype(dtype).name,
                                  "persistent_kernel_allowed": persist,
                                  "kernel": "k_pd_persist (one launch)"
                                  if ops.pd_persist_launches() > before
                                  else "k_pd_fused (one launch per iteration)",
                                  "run_ms": round(best * 1e3, 3),
                                  "us_per_iteration": round(best * 1e6 / iters, 2),
                                  "it_per_s": round(iters / best, 1),
                                  "rel_l2_vs_reference": float(err),
                                  "execution": s.get_execution()}), flush=True)
    ops.PD_PERSIST = True


if __name__ == "__main__":
    main()
