"""Pins the CPU oracle (oracle/nsol_oracle.py) against golden vectors that the
reference itself produced (tools/make_goldens.py).  CPU only."""
import numpy as np
import pytest

from oracle import nsol_oracle as orc
from conftest import rel_l2

TIGHT = 1e-12


def _zshape(shape):
    return (len(shape) * shape[0],) + tuple(shape[1:]) if len(shape) > 1 \
        else tuple(shape)


@pytest.mark.parametrize("k", ["1d", "2d", "3d"])
@pytest.mark.parametrize("tag", ["unit", "sp"])
def test_grad_and_adjoint_match_reference(golden, k, tag):
    g = golden("ops")
    sp = None if tag == "unit" else g["spacing_" + k]
    x, p = g["x_" + k], g["p_" + k]
    assert np.array_equal(orc.grad(x, sp), g["grad_%s_%s" % (k, tag)])
    assert np.allclose(orc.grad_adj(p, sp), g["gradadj_%s_%s" % (k, tag)],
                       rtol=0, atol=1e-14)
    d = x.ndim
    spv = np.ones(d) if sp is None else sp
    for a, nm in enumerate(["dx", "dy", "dz"][:d]):
        ax = d - 1 - a
        assert np.array_equal(orc.d_forward(x, ax, spv[a]),
                              g["%s_%s_%s" % (nm, k, tag)])
        assert np.array_equal(orc.d_forward_adj(x, ax, spv[a]),
                              g["%sadj_%s_%s" % (nm, k, tag)])


def test_gaussian_taps_match_reference(golden):
    g = golden("ops")
    assert np.allclose(orc.gaussian_taps(1, 2.0), g["taps_1d"], atol=1e-16)
    assert np.allclose(orc.gaussian_taps(1, 2.0, 1.5, 4), g["taps_1d_sp"],
                       atol=1e-16)
    assert np.allclose(orc.gaussian_taps(2, np.diag([2., 2.])), g["taps_2d"],
                       atol=1e-16)
    assert np.allclose(orc.gaussian_taps(2, g["cov_2d_aniso"]),
                       g["taps_2d_aniso"], atol=1e-16)
    assert np.allclose(orc.gaussian_taps(2, g["cov_2d_full"]),
                       g["taps_2d_full"], atol=1e-16)
    assert np.allclose(orc.gaussian_taps(3, np.diag([2., 2., 2.])),
                       g["taps_3d"], atol=1e-17)
    t = orc.gaussian_taps(3, g["cov_3d_aniso"])
    assert t.shape == g["taps_3d_aniso"].shape
    assert np.allclose(t, g["taps_3d_aniso"], atol=1e-17)
    assert np.allclose(
        orc.gaussian_taps(3, np.diag([4., 4., 4.]), g["spacing_3d"], 2),
        g["taps_3d_sp"], atol=1e-17)


def test_blur_matches_reference(golden):
    g = golden("ops")
    assert rel_l2(orc.gaussian_blur(g["x_1d"], 2.0), g["blur_1d"]) < TIGHT
    assert rel_l2(orc.gaussian_blur(g["x_2d"], np.diag([2., 2.])),
                  g["blur_2d"]) < TIGHT
    assert rel_l2(orc.gaussian_blur(g["x_2d"], g["cov_2d_aniso"]),
                  g["blur_2d_aniso"]) < TIGHT
    assert rel_l2(orc.gaussian_blur(g["x_2d"], g["cov_2d_full"]),
                  g["blur_2d_full"]) < TIGHT
    # axis 0 of x_3d has 7 samples but 11 taps: multiple wrap-around
    assert rel_l2(orc.gaussian_blur(g["x_3d"], np.diag([2., 2., 2.])),
                  g["blur_3d"]) < TIGHT
    assert rel_l2(orc.gaussian_blur(g["x_3d_b"], g["cov_3d_aniso"]),
                  g["blur_3d_aniso"]) < TIGHT
    assert rel_l2(orc.gaussian_blur(g["x_3d_b"], np.diag([4., 4., 4.]),
                                    g["spacing_3d"], 2),
                  g["blur_3d_sp"]) < TIGHT


@pytest.mark.parametrize("mode", ["wrap", "constant", "nearest", "reflect",
                                  "mirror"])
def test_user_kernel_convolution_modes(golden, mode):
    g = golden("ops")
    out = orc.convolve_nd(g["x_3d_b"], g["userker_3d"], mode)
    assert rel_l2(out, g["userconv_3d_" + mode]) < TIGHT
    if mode == "wrap":
        assert rel_l2(orc.convolve_nd(g["x_2d"], g["userker_2d"], mode),
                      g["userconv_2d_wrap"]) < TIGHT


def test_separable_factors():
    t = orc.gaussian_taps(3, np.diag([1., 4., 9.]))
    f = orc.separable_factors(t)
    assert f is not None and [v.size for v in f] == list(t.shape)
    rebuilt = np.multiply.outer(np.multiply.outer(f[0], f[1]), f[2])
    assert np.max(np.abs(rebuilt - t)) < 1e-16
    assert orc.separable_factors(
        orc.gaussian_taps(2, np.array([[2., .6], [.6, 1.]]))) is None


def test_prox_and_loss_match_reference(golden):
    g = golden("ops")
    v, b = g["prox_in"], g["prox_b"]
    assert np.array_equal(orc.prox_tv_conj(v, 0.7), g["prox_tv_conj"])
    assert np.array_equal(orc.prox_huber_conj(v, 0.7), g["prox_huber_conj"])
    assert np.array_equal(orc.prox_ell1_denoising(v, 0.3, b, 50.0),
                          g["prox_ell1"])
    assert np.array_equal(orc.prox_ell2_denoising(v, 0.3, b, 50.0),
                          g["prox_ell2"])
    for name in ("linear", "soft_l1", "huber", "cauchy", "arctan"):
        for fs in (1.0, 1.7):
            assert np.array_equal(orc.loss(name, g["loss_f2"], fs),
                                  g["loss_%s_%g" % (name, fs)])
            assert np.array_equal(orc.gradient_loss(name, g["loss_f2"], fs),
                                  g["gradloss_%s_%g" % (name, fs)])
    assert np.array_equal(orc.admm_prox_g(g["shrink_in"], 0.9, 3),
                          g["shrink_out"])


PD_CASES = [(k, alg, reg, data)
            for k in ("1d", "2d", "3d")
            for alg in ("ALG2", "ALG2_AHMOD", "ALG3")
            for reg in ("TV", "Huber")
            for data in ("L2", "L1")]


@pytest.mark.parametrize("k,alg,reg,data", PD_CASES)
def test_primal_dual_matches_reference(golden, k, alg, reg, data):
    g = golden("pd")
    obs = g["obs_" + k]
    L2 = {"1d": 4.0, "2d": 8.0, "3d": 16.0}[k]
    alpha = 0.05 if data == "L2" else 0.6
    out = orc.primal_dual_denoise(obs.flatten(), obs.shape, reg, data, alpha,
                                  25, L2, alg)
    assert rel_l2(out, g["pd_%s_%s_%s%s" % (k, alg, reg, data)]) < TIGHT


def test_primal_dual_cli_L2_and_spacing(golden):
    g = golden("pd")
    obs = g["obs_3d"]
    out = orc.primal_dual_denoise(obs.flatten(), obs.shape, "TV", "L2", 0.03,
                                  40, 8.0, "ALG2")
    assert rel_l2(out, g["pd_3d_ALG2_TVL2_L2eq8"]) < 1e-11
    out = orc.primal_dual_denoise(obs.flatten(), obs.shape, "TV", "L2", 0.05,
                                  25, 64.0, "ALG2", spacing=g["pd_spacing"])
    assert rel_l2(out, g["pd_3d_ALG2_TVL2_spacing"]) < TIGHT


@pytest.mark.parametrize("k,alg,reg,data", PD_CASES)
def test_c_oracle_matches_reference_and_numpy_restatement(golden, k, alg, reg,
                                                          data):
    """oracle/pd_oracle.c (the fast form the deep-iteration GPU tests and
    bench.py's cpu_baseline use) is held to the same reference goldens and is
    bit-identical to the NumPy restatement."""
    from oracle import c_oracle
    g = golden("pd")
    obs = g["obs_" + k]
    L2 = {"1d": 4.0, "2d": 8.0, "3d": 16.0}[k]
    alpha = 0.05 if data == "L2" else 0.6
    out = c_oracle.primal_dual_denoise(obs.flatten(), obs.shape, reg, data,
                                       alpha, 25, L2, alg)
    assert rel_l2(out, g["pd_%s_%s_%s%s" % (k, alg, reg, data)]) < TIGHT
    assert np.array_equal(out, orc.primal_dual_denoise(
        obs.flatten(), obs.shape, reg, data, alpha, 25, L2, alg))


def test_c_oracle_spacing_and_configs(golden):
    from oracle import c_oracle
    g = golden("pd")
    obs = g["obs_3d"]
    out = c_oracle.primal_dual_denoise(obs.flatten(), obs.shape, "TV", "L2",
                                       0.05, 25, 64.0, "ALG2",
                                       spacing=g["pd_spacing"])
    assert rel_l2(out, g["pd_3d_ALG2_TVL2_spacing"]) < TIGHT
    c = golden("configs")
    lena = c["lena_noise_u8"].astype(np.float64)
    out = c_oracle.primal_dual_denoise(lena.flatten(), lena.shape, "TV", "L2",
                                       0.03, 50, 8.0, "ALG2")
    assert rel_l2(out, c["cfg1_lena_TVL2_50it_L2eq8"]) < 2e-7  # f32 storage
    ph = c["phantom64"].astype(np.float64)
    out = c_oracle.primal_dual_denoise(ph.flatten(), ph.shape, "TV", "L2", 0.03,
                                       200, 16.0, "ALG2")
    assert rel_l2(out, c["cfg2_phantom_TVL2_200it_L2eq16"]) < 2e-7


def test_refstyle_iteration_equals_restatement(golden):
    obs = golden("pd")["obs_3d"]
    a = orc.pd_tvl2_refstyle(obs.flatten(), obs.shape, 0.05, 25, 16.0,
                             obs.max())
    assert rel_l2(a, golden("pd")["pd_3d_ALG2_TVL2"]) < TIGHT


DEC = {"1d": (50,), "2d": (18, 22), "3d": (12, 14, 16)}


def _dec_ops(golden, k):
    g = golden("admm")
    shape = DEC[k]
    cov = g["cov_" + k] if k != "1d" else float(g["cov_1d"].reshape(-1)[0])
    D, Da, A, Aa = orc.flat_operators(shape, None, cov)
    return g, shape, A, Aa, D, Da


@pytest.mark.parametrize("k", ["1d", "2d", "3d"])
def test_tikhonov_lsmr_matches_reference(golden, k):
    g, shape, A, Aa, D, Da = _dec_ops(golden, k)
    y = g["y_" + k]
    xs = float(y.max())
    I = lambda x: x.flatten()
    out = orc.tikhonov(A, Aa, I, I, y, y, alpha=0.05, x_scale=xs, iter_max=10)
    assert rel_l2(out, g["tk0_" + k]) < 1e-11
    out = orc.tikhonov(A, Aa, D, Da, y, y, alpha=0.05, x_scale=xs, iter_max=10)
    assert rel_l2(out, g["tk1_" + k]) < 1e-11
    out = orc.tikhonov(A, Aa, D, Da, y, y, alpha=0.0, x_scale=xs, iter_max=6)
    assert rel_l2(out, g["tk_noreg_" + k]) < 1e-11


@pytest.mark.parametrize("k", ["1d", "2d", "3d"])
def test_admm_lsmr_matches_reference(golden, k):
    g, shape, A, Aa, D, Da = _dec_ops(golden, k)
    y = g["y_" + k]
    out = orc.admm(A, Aa, D, Da, y, y, len(shape), alpha=0.05, rho=0.5,
                   iterations=6, iter_max=8, x_scale=float(y.max()))
    assert rel_l2(out, g["admm_lsmr_" + k]) < 1e-10


@pytest.mark.parametrize("k", ["1d", "2d"])
def test_admm_lbfgsb_huber_matches_reference(golden, k):
    g, shape, A, Aa, D, Da = _dec_ops(golden, k)
    y = g["y_" + k]
    out = orc.admm(A, Aa, D, Da, y, y, len(shape), alpha=0.05, rho=0.5,
                   iterations=3, iter_max=8, minimizer="L-BFGS-B",
                   data_loss="huber", x_scale=float(y.max()))
    assert rel_l2(out, g["admm_lbfgsb_huber_" + k]) < 1e-9


@pytest.mark.parametrize("lossname", ["huber", "soft_l1", "cauchy", "arctan",
                                      "linear"])
def test_tikhonov_minimize_losses_match_reference(golden, lossname):
    g, shape, A, Aa, D, Da = _dec_ops(golden, "2d")
    y = g["y_2d"]
    out = orc.tikhonov(A, Aa, D, Da, y, y, alpha=0.05, x_scale=float(y.max()),
                       iter_max=8, minimizer="L-BFGS-B", data_loss=lossname,
                       data_loss_scale=0.1)
    assert rel_l2(out, g["tk1_lbfgsb_%s_2d" % lossname]) < 1e-9


def _extra_ops():
    shape = (14, 18)
    D, Da, A, Aa = orc.flat_operators(shape, np.array([1.0, 2.0]),
                                      np.diag([1.5, 1.5]))
    return A, Aa, D, Da


EXTRA = [("tk_lsq_linear", dict(minimizer="lsq_linear")),
         ("tk_least_squares_linear", dict(minimizer="least_squares")),
         ("tk_least_squares_huber", dict(minimizer="least_squares",
                                         data_loss="huber",
                                         data_loss_scale=0.05)),
         ("tk_tnc_soft_l1", dict(minimizer="TNC", data_loss="soft_l1",
                                 data_loss_scale=0.1)),
         ("tk_lsmr_breg", dict(b_reg="vector")),
         ("tk_lsmr_nobounds", dict(bounds=None, x0_shift=-60.0))]


@pytest.mark.parametrize("key,kw", EXTRA)
def test_tikhonov_scipy_driver_branches(golden, key, kw):
    g = golden("extra")
    A, Aa, D, Da = _extra_ops()
    y = g["y"]
    kw = dict(kw)
    if kw.get("b_reg") == "vector":
        kw["b_reg"] = g["b_reg"]
    x0 = y + kw.pop("x0_shift", 0.0)
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = orc.tikhonov(A, Aa, D, Da, y, x0, alpha=0.05,
                               x_scale=float(y.max()), iter_max=8, **kw)
    assert rel_l2(out, g[key]) < 1e-9


def test_least_squares_huber_is_chaotic_under_1e13_perturbations(golden):
    """Documents why the GPU test holds this branch to the oracle evaluated
    with the separable blur rather than to the golden: the reference's own
    result moves by ~2e-2 when A changes by ~6e-14."""
    import warnings
    g = golden("extra")
    shape = (14, 18)
    D, Da, A, Aa = _extra_ops()[2], _extra_ops()[3], _extra_ops()[0], None
    f = orc.separable_factors(orc.gaussian_taps(
        2, np.diag([1.5, 1.5]), np.array([1.0, 2.0])))

    def A_sep(v):
        v = orc.convolve_nd(v.reshape(shape), f[0].reshape(-1, 1), "wrap")
        return orc.convolve_nd(v, f[1].reshape(1, -1), "wrap").reshape(-1)
    y = g["y"]
    assert np.max(np.abs(A_sep(y) - A(y))) < 1e-12
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = orc.tikhonov(A_sep, A_sep, D, Da, y, y, alpha=0.05,
                           x_scale=float(y.max()), iter_max=8,
                           minimizer="least_squares", data_loss="huber",
                           data_loss_scale=0.05)
    err = rel_l2(out, g["tk_least_squares_huber"])
    assert 1e-6 < err < 5e-2


def test_admm_with_vector_b_reg(golden):
    g = golden("extra")
    A, Aa, D, Da = _extra_ops()
    y = g["y"]
    out = orc.admm(A, Aa, D, Da, y, y, 2, b_reg=g["b_reg"], alpha=0.05,
                   rho=0.5, iterations=4, iter_max=6, x_scale=float(y.max()))
    assert rel_l2(out, g["admm_breg"]) < 1e-10


def test_config1_and_config2_goldens(golden):
    g = golden("configs")
    lena = g["lena_noise_u8"].astype(np.float64)
    out = orc.primal_dual_denoise(lena.flatten(), lena.shape, "TV", "L2", 0.03,
                                  50, 8.0, "ALG2")
    assert rel_l2(out, g["cfg1_lena_TVL2_50it_L2eq8"]) < 2e-7  # f32 storage
    ph = g["phantom64"].astype(np.float64)
    out = orc.primal_dual_denoise(ph.flatten(), ph.shape, "TV", "L2", 0.03,
                                  200, 16.0, "ALG2")
    assert rel_l2(out, g["cfg2_phantom_TVL2_200it_L2eq16"]) < 2e-7


def test_synth_volume_is_deterministic():
    a = orc.synth_volume(16, 0, "gauss")
    b = orc.synth_volume(16, 0, "gauss")
    assert np.array_equal(a, b) and a.shape == (16, 16, 16)
    s = orc.synth_volume(16, 3, "sp")
    assert set(np.unique(s)).issubset({0.0, 50.0, 100.0, 150.0})


# ------------------------------------------- SURVEY 8(f3) and x_scale (measures.npz)
def _xs_ops(x_gt):
    d = x_gt.ndim
    cov = 1.5 if d == 1 else np.diag(np.ones(d)) * 1.5
    D, Da, A, Aa = orc.flat_operators(x_gt.shape, None, cov)
    return A, Aa, D, Da


@pytest.mark.parametrize("k", ["1d", "2d", "3d"])
def test_prior_measures_match_reference(golden, k):
    """nsol/prior_measures.py:19-52 on the obs_* arrays of pd.npz."""
    g = golden("measures")
    obs = golden("pd")["obs_" + k]
    d = obs.ndim
    x = obs.flatten()
    assert np.isclose(orc.prior_tk0(x), g["prior_tk0_" + k], rtol=1e-14, atol=0)
    for tag, sp in (("unit", None), ("sp", g["prior_spacing_" + k])):
        D = lambda v: orc.grad(v.reshape(obs.shape), sp).reshape(-1)
        for name, val in (
                ("tk1", orc.prior_tk1(x, D)), ("tv", orc.prior_tv(x, D, d)),
                ("huber", orc.prior_huber(x, D, d)),
                ("huber_g2", orc.prior_huber(x, D, d, gamma=2.0))):
            ref = g["prior_%s_%s_%s" % (name, k, tag)]
            assert np.isclose(val, ref, rtol=1e-13, atol=0), (name, tag)


@pytest.mark.parametrize("k", ["1d", "2d"])
def test_x_scale_goldens_tikhonov_admm_pd(golden, k):
    """The set-up of tests/solvers_test.py:102-352 (Tikhonov, ADMM, PD with
    prox_linear_least_squares; x_scale = max and data divided by it)."""
    g = golden("measures")
    x_gt = g["xs_gt_1d"] if k == "1d" else g["brainweb_u8"].astype(np.float64)
    A, Aa, D, Da = _xs_ops(x_gt)
    xs = float(x_gt.max())
    for tag, s in (("unit", 1.), ("scaled", xs)):
        b = g["xs_b_%s_%s" % (k, tag)]
        out = orc.tikhonov(A, Aa, D, Da, b, b, x_scale=s)
        assert rel_l2(out, g["xs_tk_%s_%s" % (k, tag)]) < 1e-10
        out = orc.admm(A, Aa, D, Da, b, b, x_gt.ndim, x_scale=s)
        assert rel_l2(out, g["xs_admm_%s_%s" % (k, tag)]) < 1e-9
        pf = lambda x, tau: orc.prox_linear_least_squares(
            x, tau, A, Aa, b, b, x_scale=s)
        out = orc.primal_dual(pf, orc.prox_tv_conj, D, Da, 8, b, x_scale=s)
        assert rel_l2(out, g["xs_pd_%s_%s" % (k, tag)]) < 1e-9


def test_pd_deconvolution_golden(golden):
    g = golden("admm")
    shape = DEC["2d"]
    D, Da, A, Aa = orc.flat_operators(shape, None, g["cov_2d"])
    y = g["y_2d"]
    xs = float(y.max())
    pf = lambda x, tau: orc.prox_linear_least_squares(x, tau, A, Aa, y, y,
                                                      x_scale=xs)
    out = orc.primal_dual(pf, orc.prox_tv_conj, D, Da, 8, y, alpha=0.05,
                          iterations=8, x_scale=xs)
    assert rel_l2(out, g["pd_deconv_2d"]) < 1e-9


def test_similarity_identities_of_the_reference_test(golden):
    """tests/similarity_measures_test.py:20-94 on data/2D_BrainWeb.png, 4
    decimals as there (nsol.similarity_measures itself needs skimage, absent
    here: the formulas are restated from similarity_measures.py:26-120)."""
    img = golden("measures")["brainweb_u8"].astype(np.float64)
    x, x2, xo = img.flatten(), (img * 2).flatten(), (img + 2).flatten()
    assert abs(orc.sim_mae(x, xo) - np.abs(x - xo).mean()) < 1e-4
    assert abs(orc.sim_ssd(x, xo) - np.sum(np.square(x - xo))) < 1e-4
    assert abs(orc.sim_mse(x, xo) - np.square(x - xo).mean()) < 1e-4
    assert round(orc.sim_ssd(x, x), 4) == 0
    assert round(abs(orc.sim_ssd(x, xo) - x.size * 4), 4) == 0
    assert orc.sim_psnr(x, x) == np.inf
    assert round(abs(orc.sim_ncc(x, x) - 1), 4) == 0
    assert round(abs(orc.sim_ncc(x, -x) + 1), 4) == 0
    assert round(abs(orc.sim_ncc(x, xo) - 1), 4) == 0
    assert round(abs(orc.sim_ncc(x, x2) - 1), 4) == 0


def test_admm_refstyle_equals_restatement(golden):
    """The reference-style ADMM (ndimage + SciPy's own lsmr; the CPU baseline
    bench_admm.py times) against the reference's golden and the restatement."""
    g = golden("admm")
    y = g["y_3d"]
    out = orc.admm_lsmr_refstyle(y, DEC["3d"], g["cov_3d"], 0.05, 0.5, 6, 8,
                                 float(y.max()))
    assert rel_l2(out, g["admm_lsmr_3d"]) < 1e-10


CFG4_TK = [(b, "edge", it) for b in ("grad", "ident") for it in (10, 20, 32)] + \
    [("grad", "cfg4", 32), ("ident", "cfg4", 32)]


@pytest.mark.parametrize("bname,wname,iters", CFG4_TK)
def test_lsmr_at_the_normal_equations_guard_matches_reference(golden, bname,
                                                              wname, iters):
    """tests/golden/cfg4.npz: the reference's LSMR (tikhonov_linear_solver.py:
    146-158) on config 4's blur at 32^3 with the regulariser's weight where the
    build's normal-equations form starts (0.1 relative) -- the restated
    Golub-Kahan LSMR must reproduce what SciPy's produced, up to 32 iterations."""
    g = golden("cfg4")
    n = 32
    D, Da, A, _ = orc.flat_operators((n, n, n), None, np.diag([4.0, 4.0, 4.0]))
    ident = lambda v: v.reshape(-1)
    B, Ba = (D, Da) if bname == "grad" else (ident, ident)
    y = g["y_32"]
    weight = 0.1 * float(g["ratio_32"]) if wname == "edge" else 0.1
    out = orc.tikhonov(A, A, B, Ba, y, y, alpha=weight, iter_max=iters,
                       x_scale=float(y.max()))
    assert rel_l2(out, g["tk_%s_%s_%d" % (bname, wname, iters)]) < 1e-10


@pytest.mark.parametrize("bname", ["grad", "ident"])
@pytest.mark.parametrize("wname,rel", [("w005", 0.05), ("w002", 0.02)])
def test_lsmr_with_weak_regularisers_matches_reference(golden, bname, wname, rel):
    """The same blur with the regulariser at 0.05 / 0.02 of ||A g||^2 / ||g||^2 and 20
    iterations -- where a float32 LSMR is furthest from the reference and the build
    promotes the solve to float64.  (1e-8: between the dominant eigenvalues' convergence
    and its own an iterate depends on the recurrence's rounding at the 1e-9 level.)"""
    g = golden("cfg4")
    n = 32
    D, Da, A, _ = orc.flat_operators((n, n, n), None, np.diag([4.0, 4.0, 4.0]))
    ident = lambda v: v.reshape(-1)
    B, Ba = (D, Da) if bname == "grad" else (ident, ident)
    y = g["y_32"]
    out = orc.tikhonov(A, A, B, Ba, y, y, alpha=rel * float(g["ratio_32"]), iter_max=20,
                       x_scale=float(y.max()))
    assert rel_l2(out, g["tk_%s_%s_20" % (bname, wname)]) < 1e-8
