"""TVL1 / TVL2 / HuberL1 / HuberL2 denoising from the command line
(nsol_run_denoising of the reference, nsol/application/run_denoising.py:33-195,
without plotting).

    python -m nsol_amd.application.run_denoising --observation in.nii.gz \\
        --result out.nii.gz --reconstruction-type TVL2 --alpha 0.03 \\
        --iterations 50 [--reference gt.nii.gz] [--L2 8] [--dtype float32]
"""
import argparse
import sys

import numpy as np

from .. import linear_operators as LinearOperators
from .. import primal_dual_solver as pd
from .. import data_reader as dr
from .. import data_writer as dw
from .. import observer as Observer
from ..proximal_operators import ProximalOperators as prox
from ..similarity_measures import SimilarityMeasures


def build_solver(observed_nda, reconstruction_type, alpha, iterations, L2=8,
                 verbose=0, dtype=None, alg_type="ALG2"):
    """Wiring of run_denoising.py:95-154."""
    dimension = observed_nda.ndim
    b = observed_nda.flatten()
    x0 = observed_nda.flatten()
    x_scale = np.max(observed_nda)
    linear_operators = getattr(
        LinearOperators, "LinearOperators%dD" % dimension)()
    grad, grad_adj = linear_operators.get_gradient_operators()
    X_shape = observed_nda.shape
    Z_shape = (dimension * X_shape[0],) + tuple(X_shape[1:]) \
        if dimension > 1 else X_shape
    D_1D = lambda x: grad(x.reshape(*X_shape)).flatten()
    D_adj_1D = lambda x: grad_adj(x.reshape(*Z_shape)).flatten()
    if reconstruction_type in ("TVL1", "HuberL1"):
        prox_f = lambda x, tau: prox.prox_ell1_denoising(
            x, tau, x0=b, x_scale=x_scale)
    elif reconstruction_type in ("TVL2", "HuberL2"):
        prox_f = lambda x, tau: prox.prox_ell2_denoising(
            x, tau, x0=b, x_scale=x_scale)
    else:
        raise ValueError("Denoising type '%s' not known" %
                         reconstruction_type)
    prox_g_conj = prox.prox_huber_conj \
        if reconstruction_type.startswith("Huber") else prox.prox_tv_conj
    return pd.PrimalDualSolver(
        prox_f=prox_f, prox_g_conj=prox_g_conj, B=D_1D, B_conj=D_adj_1D,
        L2=L2, x0=x0, alpha=alpha, iterations=iterations, x_scale=x_scale,
        verbose=verbose, alg_type=alg_type, dtype=dtype)


def main(argv=None):
    ap = argparse.ArgumentParser(
        description="Run TVL1/TVL2/HuberL1/HuberL2 denoising on an MI355X")
    ap.add_argument("--observation", required=True)
    ap.add_argument("--result", required=False)
    ap.add_argument("--reference", required=False)
    ap.add_argument("--reconstruction-type", default="TVL2",
                    choices=["TVL1", "TVL2", "HuberL1", "HuberL2"])
    ap.add_argument("--measures", nargs="+",
                    default=["PSNR", "RMSE", "NCC"])
    ap.add_argument("--iterations", type=int, default=50)
    ap.add_argument("--solver", default="PD", choices=["PD"])
    ap.add_argument("--alpha", type=float, nargs="+", default=[0.03])
    ap.add_argument("--L2", type=float, default=8,
                    help="the reference hard-codes 8 (run_denoising.py:147); "
                         "3-D data needs >= 12 for a convergent step size")
    ap.add_argument("--alg-type", default="ALG2")
    ap.add_argument("--dtype", default="float32",
                    choices=["float32", "float64"])
    ap.add_argument("--verbose", type=int, default=0)
    args = ap.parse_args(argv)

    if len(args.alpha) == 1 and args.result is None:
        raise IOError("'--result' must be specified")

    reader = dr.DataReader(args.observation)
    reader.read_data()
    observed_nda = reader.get_data()
    x_ref = None
    if args.reference is not None:
        ref_reader = dr.DataReader(args.reference)
        ref_reader.read_data()
        x_ref = ref_reader.get_data().flatten()

    for alpha in args.alpha:
        solver = build_solver(observed_nda, args.reconstruction_type, alpha,
                              args.iterations, L2=args.L2,
                              verbose=args.verbose,
                              dtype=np.dtype(args.dtype).type,
                              alg_type=args.alg_type)
        obs = None
        if x_ref is not None:
            obs = Observer.Observer()
            obs.set_measures({
                m: (lambda x, m=m:
                    SimilarityMeasures.similarity_measures[m](x, x_ref))
                for m in args.measures})
        solver.set_observer(obs)
        solver.run()
        recon = np.array(solver.get_x().reshape(*observed_nda.shape))
        print("%s alpha=%g: %d iterations in %s (%s)" % (
            args.reconstruction_type, alpha, args.iterations,
            solver.get_computational_time(), solver.get_execution()))
        if obs is not None:
            obs.compute_measures()
            for m, vals in obs.get_measures().items():
                print("  %s: %.6g -> %.6g" % (m, vals[0], vals[-1]))
        if args.result is not None:
            dw.DataWriter(recon, args.result,
                          reader.get_image_sitk()).write_data()
    return 0


if __name__ == "__main__":
    sys.exit(main())
