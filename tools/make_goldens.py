#!/usr/bin/env python3
"""Generate golden vectors by importing the reference (gift-surg/NSoL) in the
build container.  Run:  python tools/make_goldens.py [--ref /root/reference]

Only DATA is written (inputs + the reference's outputs) into tests/golden/.
The reference source never enters this repository.  The reference's solver
modules import `pysitk.python_helper` purely for timing/printing
(solver.py:152-161, primal_dual_solver.py:238, admm_linear_solver.py:177-180);
pysitk is not installed here and cannot be fetched, so a no-arithmetic shim with
those five functions is created in a temporary directory for the duration of
this script.
"""
import argparse
import gzip
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")


def _install_print_timing_shim(tmp):
    pk = os.path.join(tmp, "pysitk")
    os.makedirs(pk)
    open(os.path.join(pk, "__init__.py"), "w").close()
    with open(os.path.join(pk, "definitions.py"), "w") as f:
        f.write("DIR_TMP = %r\n" % tmp)
    with open(os.path.join(pk, "python_helper.py"), "w") as f:
        f.write(
            "import time, datetime\n"
            "def start_timing():\n    return time.time()\n"
            "def stop_timing(t0):\n"
            "    return datetime.timedelta(seconds=time.time() - t0)\n"
            "def print_info(*a, **k):\n    pass\n"
            "def print_title(*a, **k):\n    pass\n"
            "def print_subtitle(*a, **k):\n    pass\n")
    sys.path.insert(0, tmp)


def read_nifti_f64(path):
    """Minimal NIfTI-1 reader for the reference's 64^3 float64 phantom."""
    raw = gzip.open(path, "rb").read()
    hdr = np.frombuffer(raw[:348], dtype=np.uint8)
    dim = np.frombuffer(raw[40:56], dtype="<i2")
    datatype = np.frombuffer(raw[70:72], dtype="<i2")[0]
    vox_offset = int(np.frombuffer(raw[108:112], dtype="<f4")[0])
    assert datatype == 64 and dim[0] == 3, (datatype, dim)
    nx, ny, nz = int(dim[1]), int(dim[2]), int(dim[3])
    data = np.frombuffer(raw[vox_offset:vox_offset + 8 * nx * ny * nz],
                         dtype="<f8")
    del hdr
    return data.reshape(nz, ny, nx).copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="all", choices=["all", "extra", "measures", "cfg4"],
                    help="'extra' writes only tests/golden/extra.npz (own RNG "
                         "seed, so the other fixtures stay byte-stable)")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    sys.dont_write_bytecode = True
    tmp = tempfile.mkdtemp(prefix="nsol_goldens_")
    _install_print_timing_shim(tmp)
    sys.path.insert(0, args.ref)

    import nsol.kernels as K
    import nsol.linear_operators as LO
    import nsol.loss_functions as LF
    import nsol.primal_dual_solver as PD
    import nsol.admm_linear_solver as ADMM
    import nsol.tikhonov_linear_solver as TK
    from nsol.proximal_operators import ProximalOperators as prox

    if args.only == "extra":
        make_extra(OUT, LO, TK, ADMM, PD, prox)
        return
    if args.only == "measures":
        make_measures(OUT, args.ref, LO, TK, ADMM, PD, prox)
        return
    if args.only == "cfg4":
        make_cfg4(OUT, LO, TK, ADMM)
        return

    rng = np.random.default_rng(20261003)

    # ------------------------------------------------------------------ ops
    g = {}
    shapes = {"1d": (23,), "2d": (10, 13), "3d": (7, 10, 13)}
    spac = {"1d": 1.5, "2d": np.array([2.0, 0.5]),
            "3d": np.array([2.0, 4.0, 0.5])}
    LOs = {"1d": LO.LinearOperators1D, "2d": LO.LinearOperators2D,
           "3d": LO.LinearOperators3D}
    for k, shp in shapes.items():
        d = len(shp)
        x = rng.standard_normal(shp)
        zshape = (d * shp[0],) + shp[1:] if d > 1 else shp
        p = rng.standard_normal(zshape)
        g["x_" + k] = x
        g["p_" + k] = p
        for tag, sp in (("unit", None), ("sp", spac[k])):
            lo = LOs[k]() if sp is None else LOs[k](spacing=sp)
            gr, gra = lo.get_gradient_operators()
            g["grad_%s_%s" % (k, tag)] = gr(x)
            g["gradadj_%s_%s" % (k, tag)] = gra(p)
            if sp is not None:
                g["spacing_" + k] = np.atleast_1d(sp).astype(float)
            Dx, Dxa = lo.get_dx_operators()
            g["dx_%s_%s" % (k, tag)] = Dx(x)
            g["dxadj_%s_%s" % (k, tag)] = Dxa(x)
            if d >= 2:
                Dy, Dya = lo.get_dy_operators()
                g["dy_%s_%s" % (k, tag)] = Dy(x)
                g["dyadj_%s_%s" % (k, tag)] = Dya(x)
            if d == 3:
                Dz, Dza = lo.get_dz_operators()
                g["dz_%s_%s" % (k, tag)] = Dz(x)
                g["dzadj_%s_%s" % (k, tag)] = Dza(x)

    # Gaussian taps + blur: isotropic, anisotropic (axis quirk), spacing,
    # and an axis shorter than the tap extent (tests/kernels_test.py:46 shape)
    g["taps_1d"] = K.Kernels1D().get_gaussian(2.0)
    g["taps_1d_sp"] = K.Kernels1D(spacing=1.5).get_gaussian(2.0, alpha_cut=4)
    A, _ = LO.LinearOperators1D().get_gaussian_blurring_operators(2.0)
    g["blur_1d"] = A(g["x_1d"])
    cov2 = np.diag([2.0, 2.0])
    g["taps_2d"] = K.Kernels2D().get_gaussian(cov2)
    A, _ = LO.LinearOperators2D().get_gaussian_blurring_operators(cov2)
    g["blur_2d"] = A(g["x_2d"])
    cov2a = np.diag([1.0, 4.0])
    g["cov_2d_aniso"] = cov2a
    g["taps_2d_aniso"] = K.Kernels2D().get_gaussian(cov2a)
    A, _ = LO.LinearOperators2D().get_gaussian_blurring_operators(cov2a)
    g["blur_2d_aniso"] = A(g["x_2d"])
    cov2f = np.array([[2.0, 0.6], [0.6, 1.0]])
    g["cov_2d_full"] = cov2f
    g["taps_2d_full"] = K.Kernels2D().get_gaussian(cov2f)
    A, _ = LO.LinearOperators2D().get_gaussian_blurring_operators(cov2f)
    g["blur_2d_full"] = A(g["x_2d"])
    cov3 = np.diag([2.0, 2.0, 2.0])
    g["taps_3d"] = K.Kernels3D().get_gaussian(cov3)
    A, _ = LO.LinearOperators3D().get_gaussian_blurring_operators(cov3)
    g["blur_3d"] = A(g["x_3d"])      # axis 0 has 7 < 11 taps -> multi-wrap
    cov3a = np.diag([1.0, 4.0, 9.0])
    g["cov_3d_aniso"] = cov3a
    g["taps_3d_aniso"] = K.Kernels3D().get_gaussian(cov3a)
    x3b = rng.standard_normal((20, 14, 9))
    g["x_3d_b"] = x3b
    A, _ = LO.LinearOperators3D().get_gaussian_blurring_operators(cov3a)
    g["blur_3d_aniso"] = A(x3b)
    g["taps_3d_sp"] = K.Kernels3D(spacing=spac["3d"]).get_gaussian(
        cov3 * 2, alpha_cut=2)
    A, _ = LO.LinearOperators3D(spacing=spac["3d"]).\
        get_gaussian_blurring_operators(cov3 * 2, alpha_cut=2)
    g["blur_3d_sp"] = A(x3b)
    # generic convolution with an asymmetric, even-sized user kernel
    ker = rng.standard_normal((2, 3, 4))
    g["userker_3d"] = ker
    for mode in ("wrap", "constant", "nearest", "reflect", "mirror"):
        C, Ca = LO.LinearOperators3D().\
            get_convolution_and_adjoint_convolution_operators(ker, mode=mode)
        g["userconv_3d_" + mode] = C(x3b)
    ker2 = rng.standard_normal((3, 2))
    g["userker_2d"] = ker2
    C, _ = LO.LinearOperators2D().\
        get_convolution_and_adjoint_convolution_operators(ker2)
    g["userconv_2d_wrap"] = C(g["x_2d"])

    # proxes (element-wise)
    v = 3.0 * rng.standard_normal(501)
    v[:5] = [0.0, 1.0, -1.0, 1.0 + 1e-12, -0.5]
    b0 = 100.0 * rng.random(501)
    g["prox_in"] = v
    g["prox_b"] = b0
    g["prox_tv_conj"] = prox.prox_tv_conj(np.array(v), 0.7)
    g["prox_huber_conj"] = prox.prox_huber_conj(np.array(v), 0.7)
    g["prox_ell1"] = prox.prox_ell1_denoising(np.array(v), 0.3, b0, 50.0)
    g["prox_ell2"] = prox.prox_ell2_denoising(np.array(v), 0.3, b0, 50.0)
    # losses
    f2 = np.concatenate(([0.0, 1.345 ** 2, 1.8], 9.0 * rng.random(60)))
    g["loss_f2"] = f2
    for name in ("linear", "soft_l1", "huber", "cauchy", "arctan"):
        for fs in (1.0, 1.7):
            g["loss_%s_%g" % (name, fs)] = LF.LossFunctions.get_loss[name](
                f2=f2, f_scale=fs)
            g["gradloss_%s_%g" % (name, fs)] = \
                LF.LossFunctions.get_gradient_loss[name](f2=f2, f_scale=fs)
    # ADMM isotropic shrink
    t3 = rng.standard_normal((3 * 6, 5, 4))
    g["shrink_in"] = t3
    dummy = ADMM.ADMMLinearSolver(A=None, A_adj=None, b=np.zeros(1), B=None,
                                  B_adj=None, x0=np.zeros(1), dimension=3)
    g["shrink_out"] = dummy._prox_g(t3, tau=0.9, dimension=3)
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **g)

    # ------------------------------------------------------------ PD solver
    def run_pd(obs, reg, data, alpha, iters, L2, alg, spacing=None):
        d = obs.ndim
        b = obs.flatten()
        x0 = obs.flatten()
        x_scale = np.max(obs)
        lo = LOs["%dd" % d]() if spacing is None else \
            LOs["%dd" % d](spacing=spacing)
        grad, grad_adj = lo.get_gradient_operators()
        X_shape = obs.shape
        Z_shape = grad(obs).shape
        D = lambda x: grad(x.reshape(*X_shape)).flatten()
        D_adj = lambda x: grad_adj(x.reshape(*Z_shape)).flatten()
        if data == "L1":
            pf = lambda x, tau: prox.prox_ell1_denoising(
                x, tau, x0=b, x_scale=x_scale)
        else:
            pf = lambda x, tau: prox.prox_ell2_denoising(
                x, tau, x0=b, x_scale=x_scale)
        pg = prox.prox_huber_conj if reg == "Huber" else prox.prox_tv_conj
        s = PD.PrimalDualSolver(prox_f=pf, prox_g_conj=pg, B=D, B_conj=D_adj,
                                L2=L2, x0=x0, alpha=alpha, iterations=iters,
                                x_scale=x_scale, alg_type=alg)
        s.run()
        return s.get_x()

    g = {}
    def noisy_blocks(shape):
        idx = np.indices(shape)
        v = 100.0 * ((sum(i // 4 for i in idx)) % 2) + 20.0
        return v + 8.0 * rng.standard_normal(shape)
    obs = {"1d": noisy_blocks((50,)), "2d": noisy_blocks((32, 40)),
           "3d": noisy_blocks((16, 20, 24))}
    for k, o in obs.items():
        g["obs_" + k] = o
    cases = []
    for k in ("1d", "2d", "3d"):
        L2 = {"1d": 4.0, "2d": 8.0, "3d": 16.0}[k]
        for alg in ("ALG2", "ALG2_AHMOD", "ALG3"):
            for reg in ("TV", "Huber"):
                for data in ("L2", "L1"):
                    alpha = 0.05 if data == "L2" else 0.6
                    name = "pd_%s_%s_%s%s" % (k, alg, reg, data)
                    g[name] = run_pd(obs[k], reg, data, alpha, 25, L2, alg)
                    cases.append(name)
    # CLI-faithful L2=8 on 3D (run_denoising.py:147) and non-unit spacing
    g["pd_3d_ALG2_TVL2_L2eq8"] = run_pd(obs["3d"], "TV", "L2", 0.03, 40, 8.0,
                                        "ALG2")
    g["pd_3d_ALG2_TVL2_spacing"] = run_pd(
        obs["3d"], "TV", "L2", 0.05, 25, 64.0, "ALG2",
        spacing=np.array([2.0, 4.0, 0.5]))
    g["pd_spacing"] = np.array([2.0, 4.0, 0.5])
    np.savez_compressed(os.path.join(OUT, "pd.npz"), **g)

    # ----------------------------------------------- Tikhonov / ADMM solvers
    g = {}

    def ops_for(shape, cov, spacing=None):
        d = len(shape)
        lo = LOs["%dd" % d]() if spacing is None else \
            LOs["%dd" % d](spacing=spacing)
        A, A_adj = lo.get_gaussian_blurring_operators(cov)
        grad, grad_adj = lo.get_gradient_operators()
        X = shape
        Z = grad(np.zeros(shape)).shape
        A_ = lambda x: A(x.reshape(*X)).flatten()
        Aa_ = lambda x: A_adj(x.reshape(*X)).flatten()
        D_ = lambda x: grad(x.reshape(*X)).flatten()
        Da_ = lambda x: grad_adj(x.reshape(*Z)).flatten()
        return A_, Aa_, D_, Da_

    def blocks(shape):
        idx = np.indices(shape)
        return 100.0 * ((sum(i // 4 for i in idx)) % 2) + 20.0

    dec = {"1d": ((50,), 1.5), "2d": ((18, 22), np.diag([1.5, 1.5])),
           "3d": ((12, 14, 16), np.diag([1.0, 1.0, 1.0]))}
    for k, (shape, cov) in dec.items():
        A_, Aa_, D_, Da_ = ops_for(shape, cov)
        gt = blocks(shape)
        y = A_(gt.flatten()) + 2.0 * rng.standard_normal(gt.size)
        g["gt_" + k] = gt
        g["y_" + k] = y
        g["cov_" + k] = np.atleast_2d(cov)
        xs = float(y.max())
        I_ = lambda x: x.flatten()
        # stand-alone Tikhonov TK0 / TK1 (interface :217-253)
        s = TK.TikhonovLinearSolver(A=A_, A_adj=Aa_, B=I_, B_adj=I_, b=y,
                                    x0=y, alpha=0.05, x_scale=xs, iter_max=10)
        s.run()
        g["tk0_" + k] = s.get_x()
        s = TK.TikhonovLinearSolver(A=A_, A_adj=Aa_, B=D_, B_adj=Da_, b=y,
                                    x0=y, alpha=0.05, x_scale=xs, iter_max=10)
        s.run()
        g["tk1_" + k] = s.get_x()
        # alpha below EPS -> un-augmented system (tikhonov :229, :244-248)
        s = TK.TikhonovLinearSolver(A=A_, A_adj=Aa_, B=D_, B_adj=Da_, b=y,
                                    x0=y, alpha=0.0, x_scale=xs, iter_max=6)
        s.run()
        g["tk_noreg_" + k] = s.get_x()
        # ADMM, lsmr path (interface :282-299 wiring)
        s = ADMM.ADMMLinearSolver(A=A_, A_adj=Aa_, b=y, B=D_, B_adj=Da_,
                                  x0=y, dimension=len(shape), alpha=0.05,
                                  rho=0.5, iterations=6, iter_max=8,
                                  x_scale=xs)
        s.run()
        g["admm_lsmr_" + k] = s.get_x()
        # ADMM, robust loss via L-BFGS-B (b_reg ignored by the reference)
        s = ADMM.ADMMLinearSolver(A=A_, A_adj=Aa_, b=y, B=D_, B_adj=Da_,
                                  x0=y, dimension=len(shape), alpha=0.05,
                                  rho=0.5, iterations=3, iter_max=8,
                                  minimizer="L-BFGS-B", data_loss="huber",
                                  x_scale=xs)
        s.run()
        g["admm_lbfgsb_huber_" + k] = s.get_x()
        # Tikhonov robust losses through minimize
        for lossname in ("huber", "soft_l1", "cauchy", "arctan", "linear"):
            s = TK.TikhonovLinearSolver(A=A_, A_adj=Aa_, B=D_, B_adj=Da_, b=y,
                                        x0=y, alpha=0.05, x_scale=xs,
                                        iter_max=8, minimizer="L-BFGS-B",
                                        data_loss=lossname,
                                        data_loss_scale=0.1)
            s.run()
            g["tk1_lbfgsb_%s_%s" % (lossname, k)] = s.get_x()
    # PD deconvolution (prox_linear_least_squares; interface :257-280)
    shape, cov = dec["2d"]
    A_, Aa_, D_, Da_ = ops_for(shape, cov)
    y = g["y_2d"]
    xs = float(y.max())
    pf = lambda x, tau: prox.prox_linear_least_squares(
        x=x, tau=tau, A=A_, A_adj=Aa_, b=y, x0=y, iter_max=10, x_scale=xs)
    s = PD.PrimalDualSolver(prox_f=pf, prox_g_conj=prox.prox_tv_conj, B=D_,
                            B_conj=Da_, L2=8, alpha=0.05, x0=y, iterations=8,
                            x_scale=xs)
    s.run()
    g["pd_deconv_2d"] = s.get_x()
    np.savez_compressed(os.path.join(OUT, "admm.npz"), **g)

    # ------------------------------------ BASELINE configs 1 and 2 (data/)
    g = {}
    from PIL import Image
    lena = np.array(Image.open(os.path.join(
        args.ref, "data", "2D_Lena_256_noise.png")))
    assert lena.ndim == 2
    g["lena_noise_u8"] = lena.astype(np.uint8)
    g["cfg1_lena_TVL2_50it_L2eq8"] = run_pd(
        lena.astype(np.float64), "TV", "L2", 0.03, 50, 8.0, "ALG2"
    ).astype(np.float32)
    ph3 = read_nifti_f64(os.path.join(
        args.ref, "data", "3D_SheppLoganPhantom_64.nii.gz"))
    g["phantom64"] = ph3.astype(np.float32)
    assert np.array_equal(g["phantom64"].astype(np.float64), ph3)
    noisy = ph3 + 0.05 * ph3.max() * np.random.default_rng(1).\
        standard_normal(ph3.shape)
    g["phantom64_noise_seed"] = np.array(1)
    for L2 in (8.0, 16.0):
        g["cfg2_phantom_TVL2_200it_L2eq%d" % L2] = run_pd(
            ph3, "TV", "L2", 0.03, 200, L2, "ALG2").astype(np.float32)
        g["cfg2_noisy_TVL2_200it_L2eq%d" % L2] = run_pd(
            noisy, "TV", "L2", 0.03, 200, L2, "ALG2").astype(np.float32)
    g["cfg2_noisy_TVL1_200it_L2eq16"] = run_pd(
        noisy, "TV", "L1", 0.6, 200, 16.0, "ALG2").astype(np.float32)
    g["cfg2_noisy_HuberL2_200it_L2eq16"] = run_pd(
        noisy, "Huber", "L2", 0.03, 200, 16.0, "ALG2").astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "configs.npz"), **g)
    print("goldens written to", os.path.abspath(OUT))
    for f in sorted(os.listdir(OUT)):
        print("  %-16s %8.1f KiB" % (f, os.path.getsize(
            os.path.join(OUT, f)) / 1024.0))


def make_extra(out, LO, TK, ADMM, PD, prox):
    """SciPy-driver branches of TikhonovLinearSolver (lsq_linear,
    least_squares, another `minimize` method), non-zero b_reg, bounds=None."""
    rng = np.random.default_rng(777)
    g = {}
    shape = (14, 18)
    lo = LO.LinearOperators2D(spacing=np.array([1.0, 2.0]))
    A, A_adj = lo.get_gaussian_blurring_operators(np.diag([1.5, 1.5]))
    grad, grad_adj = lo.get_gradient_operators()
    Z = grad(np.zeros(shape)).shape
    A_ = lambda x: A(x.reshape(*shape)).flatten()
    Aa_ = lambda x: A_adj(x.reshape(*shape)).flatten()
    D_ = lambda x: grad(x.reshape(*shape)).flatten()
    Da_ = lambda x: grad_adj(x.reshape(*Z)).flatten()
    idx = np.indices(shape)
    gt = 100.0 * ((idx[0] // 4 + idx[1] // 4) % 2) + 20.0
    y = A_(gt.flatten()) + 2.0 * rng.standard_normal(gt.size)
    breg = 0.5 * rng.standard_normal(2 * gt.size)
    g["y"] = y
    g["b_reg"] = breg
    xs = float(y.max())

    def tk(**kw):
        base = dict(A=A_, A_adj=Aa_, B=D_, B_adj=Da_, b=y, x0=y, alpha=0.05,
                    x_scale=xs, iter_max=8)
        base.update(kw)
        s = TK.TikhonovLinearSolver(**base)
        s.run()
        return s.get_x()

    g["tk_lsq_linear"] = tk(minimizer="lsq_linear")
    g["tk_least_squares_linear"] = tk(minimizer="least_squares")
    g["tk_least_squares_huber"] = tk(minimizer="least_squares",
                                     data_loss="huber", data_loss_scale=0.05)
    g["tk_tnc_soft_l1"] = tk(minimizer="TNC", data_loss="soft_l1",
                             data_loss_scale=0.1)
    g["tk_lsmr_breg"] = tk(b_reg=breg)
    g["tk_lsmr_nobounds"] = tk(bounds=None, x0=y - 60.0)
    s = ADMM.ADMMLinearSolver(A=A_, A_adj=Aa_, b=y, B=D_, B_adj=Da_, x0=y,
                              dimension=2, b_reg=breg, alpha=0.05, rho=0.5,
                              iterations=4, iter_max=6, x_scale=xs)
    s.run()
    g["admm_breg"] = s.get_x()
    np.savez_compressed(os.path.join(out, "extra.npz"), **g)
    print("wrote extra.npz (%.1f KiB)" %
          (os.path.getsize(os.path.join(out, "extra.npz")) / 1024.0))


def make_cfg4(out, LO, TK, ADMM):
    """BASELINE config 4 at its contract depth, at a size the reference finishes
    in a minute: sigma = 2 blur (13 taps per axis, wrap), ADMMLinearSolver with
    rho = 0.1, alpha = 0.01, 10 ADMM x 10 inner iterations
    (admm_linear_solver.py:165-218), (i) lsmr / linear and (ii) L-BFGS-B / Huber,
    on A(synth_volume(40, 0, 'clean')) + 2 % noise (the
    volume generator is build-owned and imported from the package).

    And the reference's LSMR (tikhonov_linear_solver.py:146-158) where the build
    replaces SciPy's Golub-Kahan form by Lanczos on the normal equations -- at
    the edge of that form's guard: the same blur at 32^3, B = gradient and
    B = identity, the regulariser's weight exactly 0.1 x ||A g||^2 / ||g||^2
    (g = A'b: the first Lanczos vector), 10 / 20 / 32 iterations, plus the
    config-4 weight itself (0.1 absolute) at 32 iterations."""
    sys.path.insert(0, os.path.join(HERE, ".."))
    from nsol_amd.synthetic import synth_volume
    g = {}
    cov = np.diag([4.0, 4.0, 4.0])
    lo = LO.LinearOperators3D()
    A, A_adj = lo.get_gaussian_blurring_operators(cov)
    grad, grad_adj = lo.get_gradient_operators()

    def wired(n):
        X, Z = (n, n, n), (3 * n, n, n)
        return (lambda x: A(x.reshape(*X)).flatten(),
                lambda x: A_adj(x.reshape(*X)).flatten(),
                lambda x: grad(x.reshape(*X)).flatten(),
                lambda x: grad_adj(x.reshape(*Z)).flatten())

    def observation(n):
        A_ = wired(n)[0]
        y = A_(synth_volume(n, 0, "clean").flatten())
        return y + 0.02 * y.max() * np.random.default_rng(1).standard_normal(y.size)

    n = 40
    A_, Aa_, D_, Da_ = wired(n)
    y = observation(n)
    g["y_40"] = y
    xs = float(y.max())
    for tag, kw in (("lsmr", dict()),
                    ("lbfgsb_huber", dict(minimizer="L-BFGS-B", data_loss="huber",
                                          data_loss_scale=1))):
        s = ADMM.ADMMLinearSolver(A=A_, A_adj=Aa_, b=y, B=D_, B_adj=Da_, x0=y,
                                  dimension=3, alpha=0.01, rho=0.1, iterations=10,
                                  iter_max=10, x_scale=xs, **kw)
        s.run()
        g["admm_%s_40" % tag] = s.get_x()
        print("cfg4", tag, "done")

    n = 32
    A_, Aa_, D_, Da_ = wired(n)
    I_ = lambda x: x.flatten()
    y = observation(n)
    g["y_32"] = y
    xs = float(y.max())
    gvec = Aa_(y / xs)
    ratio = float(np.sum(A_(gvec) ** 2) / np.sum(gvec ** 2))
    g["ratio_32"] = np.array(ratio)
    for bname, (B_, Ba_) in (("grad", (D_, Da_)), ("ident", (I_, I_))):
        for wname, weight in (("edge", 0.1 * ratio), ("cfg4", 0.1)):
            for iters in ((10, 20, 32) if wname == "edge" else (32,)):
                s = TK.TikhonovLinearSolver(A=A_, A_adj=Aa_, B=B_, B_adj=Ba_, b=y,
                                            x0=y, alpha=weight, x_scale=xs,
                                            iter_max=iters)
                s.run()
                g["tk_%s_%s_%d" % (bname, wname, iters)] = s.get_x()
    # weak regularisers (relative weight 0.05 / 0.02) at the iteration count where a
    # float32 LSMR is furthest from these results (20): what the build's promotion of
    # such solves to float64 has to meet
    for bname, (B_, Ba_) in (("grad", (D_, Da_)), ("ident", (I_, I_))):
        for wname, rel in (("w005", 0.05), ("w002", 0.02)):
            s = TK.TikhonovLinearSolver(A=A_, A_adj=Aa_, B=B_, B_adj=Ba_, b=y, x0=y,
                                        alpha=rel * ratio, x_scale=xs, iter_max=20)
            s.run()
            g["tk_%s_%s_20" % (bname, wname)] = s.get_x()
    np.savez_compressed(os.path.join(out, "cfg4.npz"), **g)
    print("wrote cfg4.npz (%.1f KiB)" %
          (os.path.getsize(os.path.join(out, "cfg4.npz")) / 1024.0))


def make_measures(out, ref, LO, TK, ADMM, PD, prox):
    """SURVEY section 8(f3) and the x_scale property of tests/solvers_test.py:
    * nsol.prior_measures (prior_measures.py:19-52: TK0, TK1, TV, Huber) on the
      obs_* arrays of pd.npz, unit and non-unit spacing;
    * the set-up of tests/solvers_test.py:102-352 (1-D spike signal and
      data/2D_BrainWeb.png, sigma^2 = 1.5 blur, Poisson noise, seed 1) run
      through the reference's Tikhonov, ADMM and primal-dual
      (prox_linear_least_squares) solvers with x_scale = max and with data
      divided by it.  The BrainWeb image itself is stored too: the
      similarity-measure identities of tests/similarity_measures_test.py:20-94
      are stated on it (nsol.similarity_measures cannot be imported here:
      skimage is absent)."""
    from PIL import Image
    import nsol.prior_measures as PM
    import nsol.noise as Noise
    g = {}
    pd_g = np.load(os.path.join(out, "pd.npz"))
    LOs = {1: LO.LinearOperators1D, 2: LO.LinearOperators2D,
           3: LO.LinearOperators3D}
    spac = {1: 1.5, 2: np.array([2.0, 0.5]), 3: np.array([2.0, 4.0, 0.5])}
    for k in ("1d", "2d", "3d"):
        obs = pd_g["obs_" + k]
        d = obs.ndim
        for tag, sp in (("unit", None), ("sp", spac[d])):
            lo = LOs[d]() if sp is None else LOs[d](spacing=sp)
            grad, _ = lo.get_gradient_operators()
            D = lambda x: grad(x.reshape(*obs.shape)).flatten()
            x = obs.flatten()
            g["prior_tk0_%s" % k] = np.array(
                PM.PriorMeasures.zeroth_order_tikhonov(x))
            g["prior_tk1_%s_%s" % (k, tag)] = np.array(
                PM.PriorMeasures.first_order_tikhonov(x, D))
            g["prior_tv_%s_%s" % (k, tag)] = np.array(
                PM.PriorMeasures.total_variation(x, D, d))
            g["prior_huber_%s_%s" % (k, tag)] = np.array(
                PM.PriorMeasures.huber(x, D, d))
            g["prior_huber_g2_%s_%s" % (k, tag)] = np.array(
                PM.PriorMeasures.huber(x, D, d, gamma=2.0))
        if d > 1:
            g["prior_spacing_%s" % k] = np.asarray(spac[d], dtype=float)
    g["prior_spacing_1d"] = np.array([1.5])

    brain = np.array(Image.open(os.path.join(ref, "data", "2D_BrainWeb.png")))
    assert brain.ndim == 2 and brain.dtype == np.uint8
    g["brainweb_u8"] = brain
    x1 = np.ones(50) * 50
    x1[5], x1[16], x1[23], x1[30] = 10, 100, 150, 20
    g["xs_gt_1d"] = x1
    sigma2 = 1.5
    for k, x_gt in (("1d", x1), ("2d", brain.astype(np.float64))):
        d = x_gt.ndim
        lo = LOs[d]()
        A, A_adj = lo.get_gaussian_blurring_operators(
            sigma2 if d == 1 else np.diag(np.ones(d)) * sigma2)
        grad, grad_adj = lo.get_gradient_operators()
        X = x_gt.shape
        Z = grad(x_gt).shape
        A_ = lambda x: A(x.reshape(*X)).flatten()
        Aa_ = lambda x: A_adj(x.reshape(*X)).flatten()
        D_ = lambda x: grad(x.reshape(*X)).flatten()
        Da_ = lambda x: grad_adj(x.reshape(*Z)).flatten()
        x_scale = x_gt.max()
        for tag, x_, s in (("unit", x_gt / x_scale, 1), ("scaled", x_gt, x_scale)):
            noise = Noise.Noise(A_(x_), seed=1)
            noise.add_poisson_noise(noise_level=0.05)
            b = noise.get_noisy_data().flatten()
            x0 = np.array(b)
            g["xs_b_%s_%s" % (k, tag)] = b
            sol = TK.TikhonovLinearSolver(A=A_, A_adj=Aa_, B=D_, B_adj=Da_,
                                          b=b, x0=x0, x_scale=s)
            sol.run()
            g["xs_tk_%s_%s" % (k, tag)] = sol.get_x()
            sol = ADMM.ADMMLinearSolver(A=A_, A_adj=Aa_, B=D_, B_adj=Da_, b=b,
                                        x0=x0, x_scale=s, dimension=d)
            sol.run()
            g["xs_admm_%s_%s" % (k, tag)] = sol.get_x()
            sol = PD.PrimalDualSolver(
                prox_f=lambda x, tau: prox.prox_linear_least_squares(
                    x=x, tau=tau, A=A_, A_adj=Aa_, b=b, x0=x0, x_scale=s),
                prox_g_conj=prox.prox_tv_conj, B=D_, B_conj=Da_, L2=8, x0=x0,
                x_scale=s)
            sol.run()
            g["xs_pd_%s_%s" % (k, tag)] = sol.get_x()
        for nm in ("tk", "admm", "pd"):
            dev = np.linalg.norm(g["xs_%s_%s_scaled" % (nm, k)] -
                                 x_scale * g["xs_%s_%s_unit" % (nm, k)])
            print("x_scale property, reference, %s %s: %.3e" % (nm, k, dev))
    g["xs_scale_2d"] = np.array(float(brain.max()))
    np.savez_compressed(os.path.join(out, "measures.npz"), **g)
    print("wrote measures.npz (%.1f KiB)" %
          (os.path.getsize(os.path.join(out, "measures.npz")) / 1024.0))


if __name__ == "__main__":
    main()
