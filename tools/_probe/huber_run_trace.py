#!/usr/bin/env python3
"""Config 4's Huber / L-BFGS-B branch, `runs` times and nothing else, so that a
rocprofv3 --kernel-trace --stats table divided by `runs` is one run's kernel
list.  Prints the run times, the L-BFGS-B iteration / evaluation counts of every
inner solve of the last run and the Cauchy search's counters."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import nsol_amd.linear_operators as LO  # noqa: E402
import nsol_amd.admm_linear_solver as admm  # noqa: E402
from nsol_amd import lbfgsb  # noqa: E402
from nsol_amd.synthetic import synth_volume  # noqa: E402


def factory(n, minimizer="L-BFGS-B", loss="huber"):
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
    return lambda: admm.ADMMLinearSolver(
        A=A_, A_adj=Aa_, b=y, B=D_, B_adj=Da_, x0=y, dimension=3, alpha=0.01,
        rho=0.1, iterations=10, iter_max=10, minimizer=minimizer,
        data_loss=loss, x_scale=float(y.max()), dtype=np.float32)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    minimizer = sys.argv[3] if len(sys.argv) > 3 else "L-BFGS-B"
    loss = sys.argv[4] if len(sys.argv) > 4 else "huber"
    make = factory(n, minimizer, loss)
    infos = []
    orig = lbfgsb.minimize

    def spy(*a, **k):
        x, info = orig(*a, **k)
        infos.append((info["nit"], info["nfev"], info["task"][:12]))
        return x, info
    lbfgsb.minimize = spy
    for r in range(runs):
        del infos[:]
        for k in lbfgsb.STATS:
            lbfgsb.STATS[k] = 0
        s = make()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run()
        torch.cuda.synchronize()
        print("run %d: %.4f s" % (r, time.perf_counter() - t0), flush=True)
    print("solves (nit, nfev, task):", infos)
    print("sum nit %d, nfev %d" % (sum(i[0] for i in infos), sum(i[1] for i in infos)))
    print("cauchy:", lbfgsb.STATS)


if __name__ == "__main__":
    main()
