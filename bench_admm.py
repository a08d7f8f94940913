#!/usr/bin/env python3
"""Secondary benchmark (BASELINE.json configs[3]): synthetic 512^3 Gaussian-blur
deconvolution (sigma = 2, 13 taps per axis, periodic) with ADMMLinearSolver,
TK1-regularised inner problem.  Prints one JSON line with the same contract as
bench.py: `roofline` (dominant kernel, timed live with HIP events on the state
of the run) and `cpu_baseline` (the reference-style CPU path on a bounded
sample, extrapolated per voxel).

    python bench_admm.py [--size 512] [--iterations 10] [--iter-max 10]
                         [--minimizer lsmr|L-BFGS-B] [--data-loss linear|huber]

Algorithmic bytes per voxel (float32; SURVEY 8(d) per-operator figures applied
to this build's op list):
  blur A = A^T            8   (read x, write Ax; ideal single pass)
  k_lsmr_u               40   (read Av, v, u_top, u_bot[3]; write u_top, u_bot[3])
  k_lsmr_v               24   (read A^T u, u_bot[3], v; write v)
  k_lsmr_hx              28   (read hbar, x, h, v; write hbar, x, h)
  k_admm_vw              52   (read x, v[3], w[3]; write v[3], w[3], rhs[3])
One LSMR iteration = 2 blurs + u + v + hx = 108 (the contract's figure; with the
blur's epilogue forming the top block of u the kernels move 100); one ADMM iteration
with LSMR(iter_max) = iter_max * 108 + set-up (copy of b 8, scaled rhs 24, its norm
16, A^T b 8, first v 24, h 8, zeroed x / hbar 8) + clip 8 + k_admm_vw 52
= iter_max * 108 + 156  (1 236 B per voxel for iter_max = 10).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench import HipEvents, HBM_PEAK_GBPS  # noqa: E402

BYTES = {"k_blur3_dma": 8, "k_lsmr_u": 40, "k_lsmr_v": 24, "k_lsmr_hx": 28,
         "k_admm_vw": 40,          # (52 when v itself is stored)
         # the normal-equations form: the blur with its sum of squares (reads the io
         # tile it overwrites), sum |grad y|^2 from one read of y, the three-term
         # Lanczos update with the regulariser's stencil and its norm
         "k_blur3_dma_epi": 12, "k_tk1_norm": 4, "k_tk1_lanczos": 16,
         # ... or the blur that takes both sums itself (no io tile, no second read of y)
         "k_blur3_dma_norms": 8,
         # ... or both halves of the Lanczos step inside the blur: (y, y_prev) -> (t, q0)
         # and (t, q0, y) -> y_new
         "k_blur3_lanczos_a": 16, "k_blur3_lanczos_b": 16,
         # ... or the lean pair: the first half is the blur with both sums (8), the
         # second (t, y, y_prev) -> y_new forms the step's K'K y itself
         "k_blur3_lanczos_b2": 16}
SETUP_BYTES = 156


# the kernels of the LSMR branch as rocprofv3 names them at 512^3 / float32 / 13 taps
PMC_NAMES = {"k_tk1_lanczos": "k_tk1_reg<float, 4, 4, false, 2>",
             "k_blur3_lanczos_a": "k_blur3_dma<float, 4, 13, 16, true, 3, false>",
             "k_blur3_lanczos_b": "k_blur3_dma<float, 4, 13, 16, true, 4, false>",
             "k_blur3_lanczos_b2": "k_blur3_dma<float, 4, 13, 16, true, 6, false>",
             "k_blur3_dma_norms": "k_blur3_dma<float, 4, 13, 16, true, 2, false>",
             "k_blur3_dma": "k_blur3_dma<float, 4, 13, 16, true, 0, false>",
             "k_blur3_dma_epi": "k_blur3_dma<float, 4, 13, 16, true, 1, false>",
             "k_wcomb": "k_wcomb<float, 4, true>",
             "k_admm_vw": "k_admm_vw<float, 4, 4, false, true>",
             "k_lsmr_v": "k_lsmr_v<float, 4, 4, false>",
             "k_lsmr_u": "k_lsmr_u<float, 4, 4, false>"}


def profiled_traffic(kernel, size):
    """HBM bytes per launch of one of the branch's kernels from the committed rocprofv3
    PMC passes over this script (profiles/<tag>_admm_pmc.jsonl, tools/battery.sh:
    separate FETCH_SIZE / WRITE_SIZE passes, 2 * FETCH + WRITE KiB per dispatch as
    MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be read from inside
    this process: the figure belongs to ANOTHER run of the same kernel on the same
    problem size.  Returns (bytes or None, source or None)."""
    import glob
    name = PMC_NAMES.get(kernel)
    # (the stable name first: tools/install_profiles.sh keeps a copy of the newest
    # battery's table there, so the citation outlives the renaming of a battery's files)
    stable = os.path.join(ROOT, "profiles", "latest_admm_pmc.jsonl")
    files = [stable] if os.path.exists(stable) else \
        sorted(glob.glob(os.path.join(ROOT, "profiles", "*_admm_pmc.jsonl")))
    if size != 512 or name is None or not files:
        return None, None
    fetch = write = None
    try:
        for line in open(files[-1]):
            rec = json.loads(line)
            if rec.get("kernel") == name:
                if rec["counter"] == "FETCH_SIZE":
                    fetch = float(rec["mean"])
                elif rec["counter"] == "WRITE_SIZE":
                    write = float(rec["mean"])
    except Exception:
        return None, None
    if fetch is None or write is None:
        return None, None
    return (2.0 * fetch + write) * 1024.0, \
        "from_profile: profiles/%s[%s] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes " \
        "of another run, same kernel and size; mean over its dispatches)" % (
            os.path.basename(files[-1]), name)


def bytes_per_admm_iteration(iter_max):
    return iter_max * 108 + SETUP_BYTES


def bytes_moved_per_admm_iteration(iter_max, blur_epilogue, prescaled_rhs,
                                   deferred_x=True, normal_equations=False,
                                   blur_norms=False):
    """What the kernels have to move at least with this build's fusions: the
    blur's epilogue saves the write and the read of A v (8 B per LSMR iteration),
    the pre-scaled right-hand side the scaling pass, its norm pass and the norm
    pass over b (40 B per ADMM iteration); with every v_k kept, the h / hbar / x
    update (28 B per LSMR iteration) is replaced by one pass that reads the
    iter_max + 1 stored vectors and writes x."""
    if normal_equations:
        # per Lanczos step: t = A y (12), sum |grad y|^2 (4), A^T t (8), the three-term
        # update with sa^2 grad^T grad y (16); the right-hand side A^T b + sa B^T c
        # once (8 + 24); x from the iter_max stored vectors; the outer step as before
        # (blur_norms: t = A y with both sums in 8; "lean": the second half reads t, y,
        # y_prev and writes y_new -- 24 per step, no q0)
        per_step = 24 if blur_norms == "lean" else (32 if blur_norms else 40)
        return iter_max * per_step + 32 + 4 * (iter_max + 1) + \
            SETUP_BYTES - 64 - (40 if prescaled_rhs else 0)
    per_it = (100 if blur_epilogue else 108) - (28 if deferred_x else 0)
    return iter_max * per_it + (4 * (iter_max + 2) if deferred_x else 0) + \
        SETUP_BYTES - (40 if prescaled_rhs else 0)


def time_kernels(shape, reps=20):
    """Average launch duration of each kernel of the LSMR branch on vectors of
    the run's size (HIP events on the launch stream, 3 warm-up launches)."""
    import torch
    from nsol_amd import ops
    n = int(np.prod(shape))
    dev = torch.device("cuda", torch.cuda.current_device())
    gen = torch.Generator(device=dev).manual_seed(5)
    r = lambda m: torch.rand(m, device=dev, generator=gen)
    v, ut, Av, h, hbar, x = r(n), r(n), r(n), r(n), r(n), r(n)
    ub, vv, ww, rhs = r(3 * n), r(3 * n), r(3 * n), r(3 * n)
    w = (1.0, 1.0, 1.0)
    ev = HipEvents()
    stream = torch.cuda.current_stream().cuda_stream
    # the wrappers read their reduction result back (.item()); the launches are
    # timed between events, the read-back falls outside
    lib_u = lambda: ops.lsmr_u_update(Av, v, ut, ub, ops.B_GRAD, shape, w,
                                      0.5, 0.1, -0.5, sync=False)
    lib_v = lambda: ops.lsmr_v_update(Av, ub, v, ops.B_GRAD, shape, w, 0.5,
                                      0.1, -0.5, sync=False)
    lib_hx = lambda: ops.lsmr_hx_update(hbar, x, h, v, -0.3, 0.2, -0.4, 0.5,
                                        sync=False)
    lib_vw = lambda: ops.admm_vw_update(x, None, ww, None, rhs, shape, w, 0.1,
                                        1.0)
    import nsol_amd.kernels as K
    taps = K.Kernels1D().get_gaussian(4.0)           # sigma = 2: 13 taps per axis
    blur_out = torch.empty_like(v)
    lib_blur = lambda: ops.corr3_wrap(v, shape, taps, taps, taps, out=blur_out)
    out = {}
    vts = [v, ut, Av, h, hbar, x, blur_out, r(n), r(n), r(n), r(n)]
    x_out = torch.empty_like(v)
    lib_x = lambda: ops.lincomb_many(vts, [0.1 * (k + 1) for k in range(11)],
                                     out=x_out)
    BYTES["k_wcomb"] = 4 * (11 + 1)
    slot = torch.zeros(1, dtype=torch.float64, device=dev)
    lib_epi = lambda: ops.corr3_wrap_axpby(v, blur_out, shape, taps, taps, taps, 1.0,
                                           0.0, result=slot)
    lib_reg = lambda: ops.tk1_grad_norm(v, shape, w, result=slot)
    slot2 = torch.zeros(2, dtype=torch.float64, device=dev)
    lib_norms = lambda: ops.corr3_wrap_norms(v, blur_out, shape, taps, taps, taps, w,
                                             slot2)
    lib_lz = lambda: ops.tk1_lanczos(h, Av, hbar, shape, w, 0.1, 0.5, -0.3, -0.2,
                                     out=x_out, result=slot)
    lb = ops.LanczosBoard(v, 4, 0.1, 0.0)
    lb.board[0:1] = 1.0
    lb.board[3:4] = 1.0
    lb.init()
    q0 = torch.empty_like(v)
    lib_la = lambda: ops.corr3_lanczos_a(v, h, blur_out, q0, shape, taps, taps, taps, lb, 1)
    lib_lb = lambda: ops.corr3_lanczos_b(blur_out, q0, v, x_out, shape, taps, taps, taps,
                                         lb, 1)
    lib_lb2 = lambda: ops.corr3_lanczos_b2(blur_out, v, h, x_out, shape, taps, taps, taps,
                                           lb, 1)
    for _ in range(30):        # (clocks up before the first timed kernel)
        lib_blur()
    for name, fn in (("k_blur3_dma", lib_blur), ("k_lsmr_u", lib_u),
                     ("k_lsmr_v", lib_v), ("k_lsmr_hx", lib_hx),
                     ("k_admm_vw", lib_vw), ("k_wcomb", lib_x),
                     ("k_blur3_dma_epi", lib_epi), ("k_tk1_norm", lib_reg),
                     ("k_tk1_lanczos", lib_lz), ("k_blur3_dma_norms", lib_norms),
                     ("k_blur3_lanczos_a", lib_la), ("k_blur3_lanczos_b", lib_lb),
                     ("k_blur3_lanczos_b2", lib_lb2)):
        for _ in range(3):
            fn()
        e0, e1 = ev.create(), ev.create()
        ev.record(e0, stream)
        for _ in range(reps):
            fn()
        ev.record(e1, stream)
        ms = ev.elapsed_ms(e0, e1) / reps
        gbps = BYTES[name] * n / (ms * 1e-3) / 1e9
        out[name] = {"bytes_per_voxel": BYTES[name], "avg_launch_ms": ms,
                     "achieved_GBps": gbps, "frac": gbps / HBM_PEAK_GBPS}
    return out


def cpu_baseline(sample_n, iter_max):
    """Reference-style CPU path (oracle.admm_lsmr_refstyle: dense ndimage blur,
    SciPy's lsmr) on a bounded sample: ONE ADMM iteration of LSMR(iter_max)."""
    from oracle import nsol_oracle as orc
    n = sample_n
    cov = np.diag([4.0, 4.0, 4.0])
    import scipy.ndimage
    clean = orc.synth_volume(n, 0, "clean")
    y = scipy.ndimage.convolve(clean, orc.gaussian_taps(3, cov),
                               mode="wrap").flatten()
    y = y + 0.02 * y.max() * np.random.default_rng(1).standard_normal(y.size)
    t0 = time.time()
    orc.admm_lsmr_refstyle(y, (n, n, n), cov, 0.01, 0.1, 1, iter_max,
                           float(y.max()))
    return 1.0 / (time.time() - t0)


def cpu_baseline_lbfgsb(sample_n, iter_max):
    """The robust-loss branch the way the reference runs it: dense ndimage blur,
    scipy.optimize.minimize(method="L-BFGS-B", maxiter=iter_max) with the Huber
    loss (oracle.admm -> oracle.tikhonov, tikhonov_linear_solver.py:197-220) on a
    bounded sample: ONE ADMM iteration."""
    import scipy.ndimage
    from oracle import nsol_oracle as orc
    n = sample_n
    shape = (n, n, n)
    cov = np.diag([4.0, 4.0, 4.0])
    taps = orc.gaussian_taps(3, cov)
    conv = scipy.ndimage.convolve
    A = lambda x: conv(x.reshape(*shape), taps, mode="wrap").flatten()
    D = lambda x: orc.grad(x.reshape(*shape)).flatten()
    Da = lambda p: orc.grad_adj(p.reshape(3 * n, n, n)).flatten()
    y = A(orc.synth_volume(n, 0, "clean"))
    y = y + 0.02 * y.max() * np.random.default_rng(1).standard_normal(y.size)
    t0 = time.time()
    orc.admm(A, A, D, Da, y, y, 3, alpha=0.01, iter_max=iter_max,
             minimizer="L-BFGS-B", data_loss="huber", rho=0.1, iterations=1,
             x_scale=float(y.max()))
    return 1.0 / (time.time() - t0)


# algorithmic bytes per voxel of the limited-memory products at c stored pairs
# (float32 vectors, one mask byte): W' v, W c + base vectors, the subspace matrix
# with the reduced gradient formed in the same pass
def lb_bytes(c):
    return {"k_mdots": 4 * (2 * c + 1) + 1, "k_wcomb": 4 * (2 * c + 2) + 1,
            "k_masked_gram_mfma": 4 * (2 * c + 4) + 1,
            # the 2c vectors, r, xcp, x, g and the mask read; xn and d written
            "k_subspace_step": 4 * (2 * c + 4) + 1 + 8}


def time_kernels_lbfgsb(shape, c=10, reps=6):
    """Average launch duration of the O(m n) kernels of the GPU-resident L-BFGS-B
    with a full memory (c stored pairs) on vectors of the run's size."""
    import torch
    from nsol_amd.lbfgsb_device import DeviceBackend
    n = int(np.prod(shape))
    n -= n % 16
    dev = torch.device("cuda", torch.cuda.current_device())
    gen = torch.Generator(device=dev).manual_seed(5)
    r = lambda: torch.rand(n, device=dev, generator=gen)
    be = DeviceBackend()
    ws, wy = [r() for _ in range(c)], [r() for _ in range(c)]
    x, g, z = r(), r() - 0.5, r()
    free = (torch.rand(n, device=dev, generator=gen) < 0.2).to(torch.int8)
    coef = list(np.linspace(0.1, 1.0, c))
    fns = {"k_mdots": lambda: be.dots(wy + ws, x, free),
           "k_wcomb": lambda: be.subspace_direction(z, ws, wy, coef, coef, 0.7, free),
           "k_masked_gram_mfma": lambda: be.masked_grams_rgrad(
               ws, wy, free, z, x, g, 0.7, coef, coef),
           "k_subspace_step": lambda: be.subspace_step(
               z, ws, wy, coef, coef, 0.7, free, x, x, g, 0.0, float("inf"))}
    ev = HipEvents()
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    for name, fn in fns.items():
        fn()
        e0, e1 = ev.create(), ev.create()
        torch.cuda.synchronize()
        ev.record(e0, stream)
        for _ in range(reps):
            fn()                       # (each reads its reduction back: in-order)
        ev.record(e1, stream)
        ms = ev.elapsed_ms(e0, e1) / reps
        bpv = lb_bytes(c)[name]
        gbps = bpv * n / (ms * 1e-3) / 1e9
        out[name] = {"bytes_per_voxel": bpv, "stored_pairs": c, "avg_launch_ms": ms,
                     "achieved_GBps": gbps, "frac": gbps / HBM_PEAK_GBPS}
    return out


def count_lbfgsb_calls(run):
    """Calls of the O(m n) products during one (untimed) run."""
    from nsol_amd.lbfgsb_device import DeviceBackend
    counts = {"k_mdots": 0, "k_wcomb": 0, "k_masked_gram_mfma": 0, "k_subspace_step": 0}
    orig = {k: getattr(DeviceBackend, k) for k in
            ("dots", "_wcomb", "masked_grams_rgrad", "subspace_step",
             "cauchy_setup_dots")}

    def wrap(name, key):
        def f(self, *a, **kw):
            counts[key] += 1
            return orig[name](self, *a, **kw)
        return f
    DeviceBackend.dots = wrap("dots", "k_mdots")
    DeviceBackend._wcomb = wrap("_wcomb", "k_wcomb")
    DeviceBackend.masked_grams_rgrad = wrap("masked_grams_rgrad", "k_masked_gram_mfma")
    DeviceBackend.subspace_step = wrap("subspace_step", "k_subspace_step")
    DeviceBackend.cauchy_setup_dots = wrap("cauchy_setup_dots", "k_mdots")
    try:
        run()
    finally:
        for k, v in orig.items():
            setattr(DeviceBackend, k, v)
    return counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--iter-max", type=int, default=10)
    ap.add_argument("--minimizer", default="lsmr")
    ap.add_argument("--data-loss", default="linear")
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-blur-epilogue", action="store_true",
                    help="LSMR's top-block update as its own pass instead of the "
                         "blur's epilogue (A/B runs)")
    ap.add_argument("--no-prescaled-rhs", action="store_true",
                    help="scale and norm passes over LSMR's lower right-hand side "
                         "per ADMM iteration instead of taking both from the outer "
                         "step (A/B runs)")
    ap.add_argument("--bidiag", action="store_true",
                    help="LSMR by Golub-Kahan bidiagonalisation of the augmented "
                         "operator (SciPy's form) instead of Lanczos on the normal "
                         "equations (A/B runs)")
    ap.add_argument("--carried-x", action="store_true",
                    help="LSMR carries h, hbar and x through every iteration "
                         "(SciPy's form) instead of keeping every v_k and "
                         "assembling x once (A/B runs)")
    ap.add_argument("--lb-separate", action="store_true",
                    help="L-BFGS-B: the subspace right-hand side and the subspace "
                         "step as passes of their own over the stored vectors "
                         "instead of riding on the Gram pass / fused (A/B runs)")
    ap.add_argument("--lb-capacity", type=int, default=0,
                    help="breakpoints per window of the L-BFGS-B Cauchy search "
                         "(0 = the backend's default; A/B runs)")
    ap.add_argument("--param", action="append", default=[],
                    help="library knob name=value (A/B runs, e.g. "
                         "corr_blur3_dma=0 for the register-window blur)")
    ap.add_argument("--set", action="append", default=[], dest="py_set",
                    help="module switch of the host code, nsol_amd.<module>.<NAME>="
                         "<int> (A/B runs, e.g. lsmr.CONCURRENT_NORM=0)")
    ap.add_argument("--cpu-sample", type=int, default=64,
                    help="edge length of the CPU baseline's volume (64: about "
                         "15 s for one ADMM iteration of LSMR(10))")
    args = ap.parse_args()

    import torch
    import nsol_amd.linear_operators as LO
    import nsol_amd.admm_linear_solver as admm
    from nsol_amd import ops
    from nsol_amd.synthetic import synth_volume

    from nsol_amd import _lib
    if args.no_blur_epilogue:
        LO.USE_BLUR_EPILOGUE = False
    if args.no_prescaled_rhs:
        admm.USE_PRESCALED_RHS = False
    if args.lb_separate:
        import nsol_amd.lbfgsb as lb_mod
        lb_mod.USE_GRAM_RHS = False
        lb_mod.FUSE_SUBSPACE_STEP = False
    if args.bidiag or args.carried_x:
        import nsol_amd.lsmr as lsmr_mod
        lsmr_mod.USE_NORMAL_EQUATIONS = False
    if args.carried_x:
        import nsol_amd.lsmr as lsmr_mod
        lsmr_mod.DEFER_X = False
    if args.lb_capacity:
        from nsol_amd.lbfgsb_device import DeviceBackend
        DeviceBackend.CAPACITY = args.lb_capacity
    for kv in args.param:
        k, v = kv.split("=")
        _lib.set_param(k, int(v))
    import importlib
    for kv in args.py_set:
        path, v = kv.split("=")
        mod, name = path.rsplit(".", 1)
        m = importlib.import_module("nsol_amd." + mod)
        if not hasattr(m, name):
            raise SystemExit("no switch %s" % path)
        setattr(m, name, type(getattr(m, name))(int(v)))
    n = args.size
    shape = (n, n, n)
    nvox = n ** 3
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

    def solver():
        return admm.ADMMLinearSolver(
            A=A_1D, A_adj=A_adj_1D, b=y, B=D_1D, B_adj=D_adj_1D, x0=y,
            dimension=3, alpha=0.01, rho=0.1, iterations=args.iterations,
            iter_max=args.iter_max, minimizer=args.minimizer,
            data_loss=args.data_loss, x_scale=x_scale, dtype=np.float32)
    for _ in range(max(1, args.repeat)):
        s = solver()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    x = s.get_x_device()
    rel_change = float((ops.norm2(ops.lincomb2(1.0, x, -1.0, y)) /
                        ops.norm2(y)))
    finite = bool(torch.isfinite(x).all().item())
    execution = s.get_execution()
    del s, x
    med = sorted(times)[len(times) // 2]
    out = {
        "metric": "ADMM iterations/sec on %d^3 fp32 TV deconvolution" % n,
        "value": args.iterations / med, "unit": "ADMM iterations/s",
        "n_gpus": 1, "steps": args.iterations, "warmup": 0,
        "ms_per_step": med / args.iterations * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "seconds_per_run": med, "runs": times,
        "statistic": "median of %d runs (the first includes one-time set-up)"
                     % len(times),
        "config": {"workload": "synth_volume(%d,0,'clean') blurred sigma=2 "
                               "(13 taps per axis, periodic) + 2%% noise; "
                               "ADMMLinearSolver alpha=0.01 rho=0.1 "
                               "dimension=3 (BASELINE config 4)" % n,
                   "iterations": args.iterations, "iter_max": args.iter_max,
                   "minimizer": args.minimizer, "data_loss": args.data_loss,
                   "knobs": args.param, "blur_epilogue": not args.no_blur_epilogue,
                   "prescaled_rhs": not args.no_prescaled_rhs,
                   "lsmr_x": "carried" if args.carried_x else "assembled at the end",
                   "lsmr_form": "bidiagonalisation" if (args.bidiag or args.carried_x)
                   else "Lanczos on the normal equations",
                   "lanczos_step": __import__("nsol_amd.lsmr", fromlist=["x"]).LAST_FORM[0],
                   "execution": execution},
        "rel_change_vs_input": rel_change, "finite": finite}
    if args.minimizer == "lsmr":
        kern = time_kernels(shape)
        import nsol_amd.lsmr as lsmr_mod
        deferred = bool(lsmr_mod.DEFER_X)
        normal = bool(lsmr_mod.USE_NORMAL_EQUATIONS) and deferred and \
            0.1 >= lsmr_mod.NE_MIN_WEIGHT[4] and args.iter_max <= lsmr_mod.NE_MAX_ITER
        # (k_wcomb: x assembled from the stored vectors; timed for 11)
        blur_norms = normal and bool(lsmr_mod.USE_BLUR_NORMS)
        in_blur = normal and bool(lsmr_mod.USE_BLUR_LANCZOS) and \
            lsmr_mod.LAST_FORM[0] == "lanczos-in-blur"
        lean = in_blur and bool(ops.LEAN_LANCZOS_HALVES)
        if in_blur:
            per_it = {"k_blur3_lanczos_a": 0 if lean else args.iter_max,
                      "k_blur3_lanczos_b": 0 if lean else args.iter_max,
                      "k_blur3_lanczos_b2": args.iter_max if lean else 0,
                      "k_blur3_dma_norms": args.iter_max if lean else 0, "k_blur3_dma": 0,
                      "k_blur3_dma_epi": 0, "k_tk1_norm": 0,
                      "k_tk1_lanczos": 0, "k_lsmr_v": 1, "k_lsmr_u": 0, "k_lsmr_hx": 0,
                      "k_admm_vw": 1, "k_wcomb": 1}
            blur_norms = "lean" if lean else True
        elif normal:
            per_it = {"k_blur3_dma": args.iter_max + 1,
                      "k_blur3_dma_epi": 0 if blur_norms else args.iter_max,
                      "k_tk1_norm": 0 if blur_norms else args.iter_max,
                      "k_blur3_dma_norms": args.iter_max if blur_norms else 0,
                      "k_tk1_lanczos": args.iter_max,
                      "k_blur3_lanczos_a": 0, "k_blur3_lanczos_b": 0,
                      "k_blur3_lanczos_b2": 0,
                      "k_lsmr_v": 1, "k_lsmr_u": 0, "k_lsmr_hx": 0, "k_admm_vw": 1,
                      "k_wcomb": 1}
        else:
            per_it = {"k_blur3_dma": 2 * args.iter_max + 1,
                      "k_lsmr_u": args.iter_max, "k_lsmr_v": args.iter_max + 1,
                      "k_lsmr_hx": 0 if deferred else args.iter_max,
                      "k_admm_vw": 1, "k_wcomb": 1 if deferred else 0,
                      "k_blur3_dma_epi": 0, "k_tk1_norm": 0, "k_tk1_lanczos": 0,
                      "k_blur3_dma_norms": 0, "k_blur3_lanczos_a": 0,
                      "k_blur3_lanczos_b": 0, "k_blur3_lanczos_b2": 0}
        for k, c in per_it.items():
            kern[k]["launches_per_admm_iteration"] = c
            kern[k]["ms_per_admm_iteration"] = c * kern[k]["avg_launch_ms"]
        dom = max(kern, key=lambda k: kern[k]["ms_per_admm_iteration"])
        run_bytes = bytes_per_admm_iteration(args.iter_max) * nvox
        run_gbps = run_bytes * args.iterations / med / 1e9
        traffic, traffic_source = profiled_traffic(dom, n)
        out["roofline"] = {
            "bound": "hbm", "kernel": dom,
            "achieved": kern[dom]["achieved_GBps"], "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": kern[dom]["frac"], "traffic": traffic,
            "traffic_source": traffic_source,
            "frac_traffic": (traffic / (kern[dom]["avg_launch_ms"] * 1e-3) / 1e9 /
                             HBM_PEAK_GBPS) if traffic else None,
            "avg_launch_ms": kern[dom]["avg_launch_ms"],
            "bytes_per_launch": BYTES[dom] * nvox,
            "kernels": kern,
            "whole_run": {
                "algorithmic_bytes_per_voxel_per_admm_iteration":
                    bytes_per_admm_iteration(args.iter_max),
                "bytes_moved_per_voxel_per_admm_iteration":
                    bytes_moved_per_admm_iteration(
                        args.iter_max, not args.no_blur_epilogue,
                        not args.no_prescaled_rhs, deferred, normal, blur_norms),
                "frac_moved": bytes_moved_per_admm_iteration(
                    args.iter_max, not args.no_blur_epilogue,
                    not args.no_prescaled_rhs, deferred, normal, blur_norms) * nvox * args.iterations / med / 1e9 /
                HBM_PEAK_GBPS,
                "achieved": run_gbps, "frac": run_gbps / HBM_PEAK_GBPS,
                "kernel_ms_per_admm_iteration_sum":
                    sum(k["ms_per_admm_iteration"] for k in kern.values())}}
    if args.minimizer == "L-BFGS-B":
        kern = time_kernels_lbfgsb(shape)
        calls = count_lbfgsb_calls(lambda: solver().run())
        for k in kern:
            kern[k]["launches_per_run"] = calls[k]
            kern[k]["ms_per_run_at_full_memory"] = calls[k] * kern[k]["avg_launch_ms"]
        dom = max(kern, key=lambda k: kern[k]["ms_per_run_at_full_memory"])
        out["roofline"] = {
            "bound": "hbm", "kernel": dom,
            "achieved": kern[dom]["achieved_GBps"], "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": kern[dom]["frac"], "traffic": None,
            "traffic_source": None, "avg_launch_ms": kern[dom]["avg_launch_ms"],
            "bytes_per_launch": kern[dom]["bytes_per_voxel"] * (nvox - nvox % 16),
            "note": "the O(m n) products of the limited-memory matrix (10 stored "
                    "pairs = 20 vectors per pass) dominate the branch; launch "
                    "counts are those of one run, durations those of a full "
                    "memory (the first iterations of every solve hold fewer pairs)",
            "kernels": kern}
    if not args.no_cpu_baseline and args.minimizer == "L-BFGS-B":
        sn = args.cpu_sample
        its = cpu_baseline_lbfgsb(sn, args.iter_max)
        out["cpu_baseline"] = {
            "value": its * (sn ** 3) / float(nvox),
            "unit": "ADMM iterations/s", "cores": 1, "kind": "port",
            "host_cores_available": os.cpu_count(),
            "sample": "oracle admm / tikhonov with scipy.optimize.minimize("
                      "L-BFGS-B, maxiter=%d), Huber loss, dense 13^3 ndimage "
                      "blur (reference op sequence), %d^3 volume, ONE ADMM "
                      "iteration, extrapolated per voxel to %d^3"
                      % (args.iter_max, sn, n)}
    if not args.no_cpu_baseline and args.minimizer == "lsmr":
        sn = args.cpu_sample
        its = cpu_baseline(sn, args.iter_max)
        out["cpu_baseline"] = {
            "value": its * (sn ** 3) / float(nvox),
            "unit": "ADMM iterations/s", "cores": 1, "kind": "port",
            "host_cores_available": os.cpu_count(),
            "sample": "oracle admm_lsmr_refstyle (dense 13^3 ndimage blur, "
                      "scipy.sparse.linalg.lsmr, reference op sequence), %d^3 "
                      "volume, ONE ADMM iteration of LSMR(%d), extrapolated "
                      "per voxel to %d^3" % (sn, args.iter_max, n)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
