"""TK0L2 / TK1L2 / TVL2 / HuberL2 deconvolution from the command line
(nsol_run_deconvolution of the reference, nsol/application/
run_deconvolution.py:28-245 and the wiring of
deconvolution_solver_parameter_study_interface.py:217-325, without plotting).

    python -m nsol_amd.application.run_deconvolution --observation blurred.png \\
        --result out.png --blur 2 --reconstruction-type TVL2 --solver ADMM
"""
import argparse
import sys

import numpy as np

from .. import linear_operators as LinearOperators
from .. import primal_dual_solver as pd
from .. import admm_linear_solver as admm
from .. import tikhonov_linear_solver as tk
from .. import data_reader as dr
from .. import data_writer as dw
from ..proximal_operators import ProximalOperators as prox


def build_solver(observed_nda, spacing, blur, reconstruction_type="TVL2",
                 tv_solver="PD", alpha=0.01, iterations=10, iter_max=10,
                 rho=0.1, minimizer="lsmr", data_loss="linear",
                 data_loss_scale=1., L2=8, verbose=0, dtype=None):
    dimension = observed_nda.ndim
    sigma = np.atleast_1d(blur).astype(float)
    cov = np.diag(np.ones(dimension)) * sigma ** 2
    if dimension == 1:
        cov = float(cov.reshape(-1)[0])
    b = observed_nda.flatten()
    x0 = observed_nda.flatten()
    x_scale = np.max(observed_nda)
    lo = getattr(LinearOperators, "LinearOperators%dD" % dimension)(
        spacing=spacing)
    A, A_adj = lo.get_gaussian_blurring_operators(cov)
    grad, grad_adj = lo.get_gradient_operators()
    X = observed_nda.shape
    Z = (dimension * X[0],) + tuple(X[1:]) if dimension > 1 else X
    A_1D = lambda x: A(x.reshape(*X)).flatten()
    A_adj_1D = lambda x: A_adj(x.reshape(*X)).flatten()
    D_1D = lambda x: grad(x.reshape(*X)).flatten()
    D_adj_1D = lambda x: grad_adj(x.reshape(*Z)).flatten()
    I_1D = lambda x: x.flatten()
    common = dict(A=A_1D, A_adj=A_adj_1D, b=b, x0=x0, alpha=alpha,
                  x_scale=x_scale, data_loss=data_loss,
                  data_loss_scale=data_loss_scale, iter_max=iter_max,
                  verbose=verbose, dtype=dtype)
    if reconstruction_type == "TK0L2":
        return tk.TikhonovLinearSolver(B=I_1D, B_adj=I_1D,
                                       minimizer=minimizer, **common)
    if reconstruction_type == "TK1L2":
        return tk.TikhonovLinearSolver(B=D_1D, B_adj=D_adj_1D,
                                       minimizer=minimizer, **common)
    if reconstruction_type == "TVL2" and tv_solver == "ADMM":
        return admm.ADMMLinearSolver(B=D_1D, B_adj=D_adj_1D, rho=rho,
                                     iterations=iterations,
                                     dimension=dimension,
                                     minimizer=minimizer, **common)
    if reconstruction_type in ("TVL2", "HuberL2"):
        prox_f = lambda x, tau: prox.prox_linear_least_squares(
            x=x, tau=tau, A=A_1D, A_adj=A_adj_1D, b=b, x0=x0,
            iter_max=iter_max, data_loss=data_loss,
            data_loss_scale=data_loss_scale, x_scale=x_scale)
        pg = prox.prox_tv_conj if reconstruction_type == "TVL2" \
            else prox.prox_huber_conj
        return pd.PrimalDualSolver(prox_f=prox_f, prox_g_conj=pg, B=D_1D,
                                   B_conj=D_adj_1D, L2=L2, alpha=alpha,
                                   x0=x0, iterations=iterations,
                                   x_scale=x_scale, verbose=verbose,
                                   dtype=dtype)
    raise ValueError("Reconstruction type '%s' not known" %
                     reconstruction_type)


def main(argv=None):
    ap = argparse.ArgumentParser(
        description="Run TK0L2/TK1L2/TVL2/HuberL2 deconvolution on an MI355X")
    ap.add_argument("--observation", required=True)
    ap.add_argument("--result", required=True)
    ap.add_argument("--blur", type=float, default=1.2)
    ap.add_argument("--reconstruction-type", default="TVL2",
                    choices=["TK0L2", "TK1L2", "TVL2", "HuberL2"])
    ap.add_argument("--solver", default="PD", choices=["PD", "ADMM"])
    ap.add_argument("--alpha", type=float, nargs="+", default=[0.01])
    ap.add_argument("--rho", type=float, default=0.1)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--iter-max", type=int, default=10)
    ap.add_argument("--minimizer", default="lsmr")
    ap.add_argument("--data-loss", default="linear")
    ap.add_argument("--data-loss-scale", type=float, default=1.)
    ap.add_argument("--L2", type=float, default=8)
    ap.add_argument("--dtype", default="float32",
                    choices=["float32", "float64"])
    ap.add_argument("--verbose", type=int, default=0)
    args = ap.parse_args(argv)

    reader = dr.DataReader(args.observation)
    reader.read_data()
    observed_nda = reader.get_data()
    info = reader.get_image_sitk()
    spacing = np.ones(observed_nda.ndim) if info is None \
        else np.array(info.GetSpacing())
    for alpha in args.alpha:
        solver = build_solver(
            observed_nda, spacing, args.blur, args.reconstruction_type,
            args.solver, alpha, args.iterations, args.iter_max, args.rho,
            args.minimizer, args.data_loss, args.data_loss_scale, args.L2,
            args.verbose, np.dtype(args.dtype).type)
        solver.run()
        recon = np.array(solver.get_x().reshape(*observed_nda.shape))
        print("%s alpha=%g: %s" % (args.reconstruction_type, alpha,
                                   solver.get_computational_time()))
        dw.DataWriter(recon, args.result, info).write_data()
    return 0


if __name__ == "__main__":
    sys.exit(main())
