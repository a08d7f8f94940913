#!/usr/bin/env python3
"""Host-side profile (cProfile) of one ADMMLinearSolver.run() after a warm-up
run: shows where the Python driver of the LSMR / L-BFGS-B paths spends time."""
import cProfile
import io
import pstats
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nsol_amd.linear_operators as LO  # noqa: E402
import nsol_amd.admm_linear_solver as admm  # noqa: E402
from nsol_amd.synthetic import synth_volume  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    minimizer = sys.argv[2] if len(sys.argv) > 2 else "L-BFGS-B"
    loss = sys.argv[3] if len(sys.argv) > 3 else "huber"
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    shape = (n, n, n)
    lo = LO.LinearOperators3D()
    A, A_adj = lo.get_gaussian_blurring_operators(np.diag([4.0, 4.0, 4.0]))
    grad, grad_adj = lo.get_gradient_operators()
    Z = (3 * n, n, n)
    A_ = lambda x: A(x.reshape(*shape)).flatten()
    Aa_ = lambda x: A_adj(x.reshape(*shape)).flatten()
    D_ = lambda x: grad(x.reshape(*shape)).flatten()
    Da_ = lambda x: grad_adj(x.reshape(*Z)).flatten()
    clean = torch.from_numpy(synth_volume(n, 0, "clean", np.float32)).cuda()
    y = A(clean).flatten()
    gen = torch.Generator(device="cuda").manual_seed(1)
    y = y + 0.02 * float(y.max()) * torch.randn(y.shape, device="cuda",
                                                 generator=gen)

    def make():
        return admm.ADMMLinearSolver(
            A=A_, A_adj=Aa_, b=y, B=D_, B_adj=Da_, x0=y, dimension=3,
            alpha=0.01, rho=0.1, iterations=iters, iter_max=10,
            minimizer=minimizer, data_loss=loss, x_scale=float(y.max()),
            dtype=np.float32)
    make().run()
    s = make()
    pr = cProfile.Profile()
    pr.enable()
    s.run()
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(22)
    print(out.getvalue()[:5000])
    print("run time:", s.get_computational_time())
    from nsol_amd import lbfgsb
    print("lbfgsb stats (both runs):", lbfgsb.STATS)


if __name__ == "__main__":
    main()
