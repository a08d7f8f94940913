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
             "k_admm_vw_g": "k_admm_vw_g<float, 4, false>",
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


# C-ABI entries as nsol_amd/_timing.py names them -> (kernel key for BYTES / PMC_NAMES,
# algorithmic bytes per voxel; a callable takes the entry's tag = number of vectors)
ENTRIES = {
    "corr3_wrap": ("k_blur3_dma", 8),
    "corr3_wrap_axpby": ("k_blur3_dma_epi", 12),
    "corr3_wrap_norms": ("k_blur3_dma_norms", 8),
    "corr3_wrap_lanczos_a": ("k_blur3_lanczos_a", 16),
    "corr3_wrap_lanczos_b": ("k_blur3_lanczos_b", 16),
    "corr3_wrap_lanczos_a2": ("k_blur3_dma_norms", 8),
    "corr3_wrap_lanczos_b2": ("k_blur3_lanczos_b2", 16),
    "corr3_wrap_loss": ("k_blur3_loss", 12),
    "tk1_grad_norm": ("k_tk1_norm", 4),
    "tk1_lanczos": ("k_tk1_lanczos", 16),
    "tk1_reg_cost_grad": ("k_tk1_cost_grad", 12),
    # reads x, g (in place), d, the old gradient; writes the gradient and its difference
    "tk1_reg_objective": ("k_tk1_objective", 24),
    "lsmr_u_update": ("k_lsmr_u", 40),
    "lsmr_v_update": ("k_lsmr_v", 24), "lsmr_v_update_to": ("k_lsmr_v", 24),
    "lsmr_hx_update": ("k_lsmr_hx", 28),
    "admm_vw_update": ("k_admm_vw", 40), "admm_vw_update_norm": ("k_admm_vw", 40),
    # the outer step and the next solve's start vector in one pass: x, w (3), A^T b read;
    # w (3), g written
    "admm_vw_update_g": ("k_admm_vw_g", 36),
    "lincomb_clip": ("k_wcomb", lambda k: 4 * (k + 1)),
    "lb_wcomb": ("k_wcomb", lambda k: 4 * (k + 2) + 1),
    "lb_mdots": ("k_mdots", lambda k: 4 * (k + 1) + 1),
    "lb_masked_gram_rgrad": ("k_masked_gram_mfma", lambda k: 4 * (k + 4) + 1),
    "lb_subspace_step": ("k_subspace_step", lambda k: 4 * (k + 4) + 1 + 8),
    "lb_subspace_step_r": ("k_subspace_step", lambda k: 4 * (k + 3) + 1 + 8),
    "lb_cauchy_setup": ("k_cauchy_setup", 18),
    "lb_cauchy_finish": ("k_cauchy_finish", 16),
    "lb_projgr": ("k_projgr", 8),
    "dot": ("k_dot", 8), "scale": ("k_map", 8), "clip": ("k_map", 8),
    "lincomb2": ("k_map", 12), "grad": ("k_grad", 16),
}
# the integer argument that says how many vectors an entry runs over
ENTRY_TAGS = {"lincomb_clip": 2, "lb_wcomb": 7, "lb_mdots": 1, "lb_masked_gram_rgrad": 1,
              "lb_subspace_step": 2, "lb_subspace_step_r": 2}


def kernels_in_run(run_once, nvox):
    """Durations of the C-ABI entries INSIDE one run of the solver (event pairs around
    every entry, nsol_amd/_timing.py): cold caches, real neighbours, the clocks of the
    moment -- what rocprofv3 sees for the same run, not a replay on idle operands.
    Returns (table by entry, wall seconds of that run)."""
    import torch
    from nsol_amd import _timing
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with _timing.KernelTimer(ENTRY_TAGS) as kt:
        run_once()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    table = {}
    for name, s in kt.summary().items():
        base, _, tag = name.partition("#")
        kern, bpv = ENTRIES.get(base, (None, None))
        if callable(bpv):
            bpv = bpv(int(tag))
        rec = {"kernel": kern, "launches_per_run": s["launches"],
               "avg_launch_ms": s["avg_ms"], "min_ms": s["min_ms"], "max_ms": s["max_ms"],
               "ms_per_run": s["total_ms"], "bytes_per_voxel": bpv}
        if bpv is not None:
            gbps = bpv * nvox / (s["avg_ms"] * 1e-3) / 1e9
            rec["achieved_GBps"], rec["frac"] = gbps, gbps / HBM_PEAK_GBPS
        table[name] = rec
    return table, wall


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
    out = measure(args.size, args.iterations, args.iter_max, args.minimizer,
                  args.data_loss, args.repeat,
                  cpu_sample=0 if args.no_cpu_baseline else args.cpu_sample)
    out["config"].update({
        "knobs": args.param, "blur_epilogue": not args.no_blur_epilogue,
        "prescaled_rhs": not args.no_prescaled_rhs,
        "lsmr_x": "carried" if args.carried_x else "assembled at the end"})
    print(json.dumps(out))


def measure(n, iterations=10, iter_max=10, minimizer="lsmr", data_loss="linear",
            repeat=5, cpu_sample=64):
    """One JSON-able record for BASELINE config 4 at edge length n: `repeat` untimed-by-
    events runs (median wall time = seconds_per_run), ONE more run with every C-ABI
    entry bracketed by events (the `roofline`: per-kernel durations as they are inside
    the solve), and the CPU path on a cpu_sample^3 volume (0: skipped).  bench.py puts
    this record into the driver's line as `config4`."""
    import torch
    import nsol_amd.linear_operators as LO
    import nsol_amd.admm_linear_solver as admm
    import nsol_amd.lsmr as lsmr_mod
    from nsol_amd import ops
    from nsol_amd.synthetic import synth_volume
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
    del clean
    gen = torch.Generator(device="cuda").manual_seed(1)
    y = y + 0.02 * float(y.max()) * torch.randn(y.shape, device="cuda",
                                                 generator=gen)
    x_scale = float(y.max())
    times = []

    def solver():
        return admm.ADMMLinearSolver(
            A=A_1D, A_adj=A_adj_1D, b=y, B=D_1D, B_adj=D_adj_1D, x0=y,
            dimension=3, alpha=0.01, rho=0.1, iterations=iterations,
            iter_max=iter_max, minimizer=minimizer,
            data_loss=data_loss, x_scale=x_scale, dtype=np.float32)
    for _ in range(max(1, repeat)):
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
    inner = s.get_inner_log()
    del s, x
    # (the first run carries the one-time set-up when there is more than one)
    med = sorted(times[1:] or times)[len(times[1:] or times) // 2]
    kern, timed_wall = kernels_in_run(lambda: solver().run(), nvox)
    known = {k: v for k, v in kern.items() if v["bytes_per_voxel"] is not None}
    dom = max(known, key=lambda k: known[k]["ms_per_run"])
    dkey = kern[dom]["kernel"]
    traffic, traffic_source = profiled_traffic(dkey, n)
    kernel_ms = sum(v["ms_per_run"] for v in kern.values())
    out = {
        "metric": "ADMM iterations/sec on %d^3 fp32 TV deconvolution" % n,
        "value": iterations / med, "unit": "ADMM iterations/s",
        "n_gpus": 1, "steps": iterations, "warmup": 0,
        "ms_per_step": med / iterations * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "seconds_per_run": med, "runs": times,
        "statistic": "median of the runs after the first (which carries the one-time "
                     "set-up)" if len(times) > 1 else "one run",
        "config": {"workload": "synth_volume(%d,0,'clean') blurred sigma=2 "
                               "(13 taps per axis, periodic) + 2%% noise; "
                               "ADMMLinearSolver alpha=0.01 rho=0.1 "
                               "dimension=3 (BASELINE config 4)" % n,
                   "iterations": iterations, "iter_max": iter_max,
                   "minimizer": minimizer, "data_loss": data_loss,
                   "lsmr_form": lsmr_mod.LAST_FORM[0] if minimizer == "lsmr" else None,
                   "execution": execution, "inner_solves": inner},
        "rel_change_vs_input": rel_change, "finite": finite,
        "roofline": {
            "bound": "hbm", "kernel": dkey, "entry": dom,
            "achieved": kern[dom]["achieved_GBps"], "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": kern[dom]["frac"], "traffic": traffic,
            "traffic_source": traffic_source,
            "frac_traffic": (traffic / (kern[dom]["avg_launch_ms"] * 1e-3) / 1e9 /
                             HBM_PEAK_GBPS) if traffic else None,
            "avg_launch_ms": kern[dom]["avg_launch_ms"],
            "launches": kern[dom]["launches_per_run"],
            "bytes_per_launch": kern[dom]["bytes_per_voxel"] * nvox,
            "how": "event pairs around every C-ABI entry of ONE run of the solver "
                   "(nsol_amd/_timing.py); the dominant entry is the one with the most "
                   "time in that run among those with stated algorithmic bytes",
            "timed_run_wall_s": timed_wall,
            "timed_run_kernel_ms": kernel_ms,
            "timed_run_gpu_idle_ms": timed_wall * 1e3 - kernel_ms,
            "kernels": kern}}
    if minimizer == "lsmr":
        deferred = bool(lsmr_mod.DEFER_X)
        form = lsmr_mod.LAST_FORM[0]
        normal = form in ("lanczos", "lanczos-in-blur")
        blur_norms = normal and bool(lsmr_mod.USE_BLUR_NORMS)
        if form == "lanczos-in-blur":
            blur_norms = "lean" if ops.LEAN_LANCZOS_HALVES else True
        prescaled = bool(admm.USE_PRESCALED_RHS)
        epilogue = bool(LO.USE_BLUR_EPILOGUE)
        run_bytes = bytes_per_admm_iteration(iter_max) * nvox
        run_gbps = run_bytes * iterations / med / 1e9
        moved = bytes_moved_per_admm_iteration(iter_max, epilogue, prescaled, deferred,
                                               normal, blur_norms)
        # the step after the last x-update is not computed (40 B once per run); with the
        # one-pass outer step (nsol_admm_vw_update_g_*) the other steps and the start
        # vector of the solve behind them move 36 B instead of 40 + 24
        one_pass = bool(admm.USE_ONE_PASS_OUTER_STEP) and prescaled and normal
        moved -= 40.0 / iterations
        if one_pass:
            moved -= 28.0 * (iterations - 1) / iterations
        out["roofline"]["whole_run"] = {
            "algorithmic_bytes_per_voxel_per_admm_iteration":
                bytes_per_admm_iteration(iter_max),
            "bytes_moved_per_voxel_per_admm_iteration": moved,
            "one_pass_outer_step": one_pass,
            "frac_moved": moved * nvox * iterations / med / 1e9 / HBM_PEAK_GBPS,
            "achieved": run_gbps, "frac": run_gbps / HBM_PEAK_GBPS}
    if cpu_sample and minimizer == "L-BFGS-B":
        sn = cpu_sample
        its = cpu_baseline_lbfgsb(sn, iter_max)
        out["cpu_baseline"] = {
            "value": its * (sn ** 3) / float(nvox),
            "unit": "ADMM iterations/s", "cores": 1, "kind": "port",
            "host_cores_available": os.cpu_count(),
            "sample": "oracle admm / tikhonov with scipy.optimize.minimize("
                      "L-BFGS-B, maxiter=%d), Huber loss, dense 13^3 ndimage "
                      "blur (reference op sequence), %d^3 volume, ONE ADMM "
                      "iteration, extrapolated per voxel to %d^3"
                      % (iter_max, sn, n)}
    if cpu_sample and minimizer == "lsmr":
        sn = cpu_sample
        its = cpu_baseline(sn, iter_max)
        out["cpu_baseline"] = {
            "value": its * (sn ** 3) / float(nvox),
            "unit": "ADMM iterations/s", "cores": 1, "kind": "port",
            "host_cores_available": os.cpu_count(),
            "sample": "oracle admm_lsmr_refstyle (dense 13^3 ndimage blur, "
                      "scipy.sparse.linalg.lsmr, reference op sequence), %d^3 "
                      "volume, ONE ADMM iteration of LSMR(%d), extrapolated "
                      "per voxel to %d^3" % (sn, iter_max, n)}
    return out


if __name__ == "__main__":
    main()
