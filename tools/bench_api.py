#!/usr/bin/env python3
"""Config 3 through the PUBLIC API (NumPy in, NumPy out): PrimalDualSolver built
from caller-side lambdas exactly as run_denoising.py does, 512^3 float32,
500 iterations.  Reports upload / run() / download separately (the
PCIe-inclusive figure that DESIGN.md quotes)."""
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
from nsol_amd.synthetic import synth_volume  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    obs = synth_volume(n, 0, "gauss", np.float32)
    X = obs.shape
    b = obs.reshape(-1)
    xs = float(obs.max())
    grad, grad_adj = LO.LinearOperators3D().get_gradient_operators()
    Z = (3 * n, n, n)
    D = lambda x: grad(x.reshape(*X)).flatten()
    Da = lambda x: grad_adj(x.reshape(*Z)).flatten()
    pf = lambda x, tau: prox.prox_ell2_denoising(x, tau, x0=b, x_scale=xs)
    torch.zeros(1, device="cuda")
    t0 = time.perf_counter()
    s = pd.PrimalDualSolver(prox_f=pf, prox_g_conj=prox.prox_tv_conj, B=D,
                            B_conj=Da, L2=16, x0=b, alpha=0.03,
                            iterations=iters, x_scale=xs)
    t1 = time.perf_counter()
    s.run()
    t2 = time.perf_counter()
    x = s.get_x()
    t3 = time.perf_counter()
    print(json.dumps({
        "size": n, "iterations": iters, "execution": s.get_execution(),
        "construct_s": round(t1 - t0, 3), "run_s_incl_upload": round(t2 - t1, 3),
        "get_x_s": round(t3 - t2, 3),
        "it_per_s_run": round(iters / (t2 - t1), 1),
        "it_per_s_end_to_end": round(iters / (t3 - t0), 1),
        "finite": bool(np.isfinite(x).all())}))


if __name__ == "__main__":
    main()
