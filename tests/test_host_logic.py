"""CPU-only tests: C-ABI library loads and exports every declared symbol, the
host-side step schedule, operator recognition through caller lambdas, tap
definitions, error behaviour, synthetic inputs and the sharded batch runner
(gloo, world_size 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_builds_loads_and_exports_declared_symbols():
    from nsol_amd.build import build_library
    from nsol_amd import _lib
    path = build_library()
    assert os.path.exists(path)
    decl = _lib.declared_symbols()
    assert len(decl) >= 40
    for base in ("grad", "grad_adj", "diff_axis", "corr_axis", "corr_dense",
                 "lincomb2", "lincomb3", "scale", "clip", "prox_dual_clamp",
                 "prox_ell1", "prox_ell2", "dot", "pd_dual_step",
                 "pd_primal_step", "pd_fused_iter", "pd_run",
                 "admm_vw_update", "admm_vw_update_g", "vector_shrink", "loss_cost_grad",
                 "loss_eval", "vector_norm_sum", "pd_fused2_iter",
                 "pd_fusedk_iter", "corr3_wrap", "lb_masked_gram", "lb_mdot",
                 "tk1_reg_cost_grad", "lb_diff_dots",
                 "loss_residual_cost_grad"):
        for suf in ("f32", "f64"):
            assert "nsol_%s_%s" % (base, suf) in decl
    lib = _lib.load()            # binds every symbol or raises
    assert lib.nsol_hip_abi_version() == 1
    assert lib.nsol_hip_reduce_ws_doubles() == 65536
    assert lib.nsol_lb_gram_ws_doubles() >= 256
    assert lib.nsol_pd_fusedk_launches(3) == 0 and lib.nsol_pd_fusedk_launches(4) == -1
    for name in ("nsol_pd_fusedk_tuned", "nsol_pd_fusedk_plan",
                 "nsol_pd_fusedk_launches"):
        assert name in decl


def test_every_tuning_knob_named_in_python_exists_in_the_library():
    """tools/, bench and tests set kernel parameters by name; a knob that was
    removed from the HIP sources must not linger in a script."""
    import glob
    import re
    csrc = "".join(open(f).read() for f in
                   glob.glob(os.path.join(ROOT, "nsol_amd", "csrc", "*.hip")))
    names = set()
    for dirpath, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "__pycache__")]
        for f in files:
            if f.endswith(".py"):
                names.update(re.findall(r'set_param\("([a-z0-9_]+)"',
                                        open(os.path.join(dirpath, f)).read()))
    assert len(names) > 10
    missing = sorted(n for n in names if '"%s"' % n not in csrc)
    assert not missing, missing
    from nsol_amd import _lib
    unknown = "pdk_no_such_knob"
    with pytest.raises(ValueError):
        _lib.set_param(unknown, 1)


def test_product_does_not_import_the_oracle():
    import re
    pkg = os.path.join(ROOT, "nsol_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt,
                                     re.M), f


def test_missing_library_fails_loudly(monkeypatch):
    from nsol_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libnsol_hip.so")
    with pytest.raises(_lib.NsolHipError):
        _lib.load()


@pytest.mark.parametrize("alg", ["ALG2", "ALG2_AHMOD", "ALG3"])
def test_step_schedule_matches_oracle(alg):
    from nsol_amd.primal_dual_solver import step_schedule
    from oracle import nsol_oracle as orc
    a = step_schedule(alg, 16.0, 1 / 0.03, 40)
    b = orc.pd_schedule(alg, 16.0, 1 / 0.03, 40)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    with pytest.raises(KeyError):
        step_schedule("ALG9", 16.0, 1.0, 3)


def test_taps_match_reference_goldens(golden):
    import nsol_amd.kernels as K
    g = golden("ops")
    assert np.array_equal(K.Kernels1D().get_gaussian(2.0), g["taps_1d"])
    assert np.array_equal(K.Kernels2D().get_gaussian(g["cov_2d_full"]),
                          g["taps_2d_full"])
    assert np.array_equal(K.Kernels3D().get_gaussian(g["cov_3d_aniso"]),
                          g["taps_3d_aniso"])
    k3 = K.Kernels3D(spacing=np.array([2., 4., 8.]))
    # tests/kernels_test.py:59-136: nonzero taps equal 1/spacing[axis]
    assert k3.get_dx_forward_difference().shape == (1, 1, 2)
    assert np.allclose(np.abs(k3.get_dx_forward_difference()), 0.5)
    assert k3.get_dy_backward_difference().shape == (1, 3, 1)
    assert np.allclose(k3.get_dy_backward_difference().ravel(),
                       [0, 0.25, -0.25])
    assert np.allclose(k3.get_dz_forward_difference().ravel(),
                       [0.125, -0.125])
    with pytest.raises(ValueError):
        K.Kernels3D(spacing=np.ones(2))
    with pytest.raises(ValueError):
        K.Kernels2D().get_gaussian(np.ones((3, 3)))


def _wired_solver(obs, data="L2", reg="TV"):
    import nsol_amd.linear_operators as LO
    import nsol_amd.primal_dual_solver as pd
    from nsol_amd.proximal_operators import ProximalOperators as prox
    from nsol_amd.symbolic import Sym
    lo = {1: LO.LinearOperators1D, 2: LO.LinearOperators2D,
          3: LO.LinearOperators3D}[obs.ndim](spacing=np.ones(obs.ndim) * 2.0)
    grad, grad_adj = lo.get_gradient_operators()
    X = obs.shape
    Z = grad(Sym(X)).shape
    b = obs.flatten()
    D = lambda x: grad(x.reshape(*X)).flatten()
    Da = lambda x: grad_adj(x.reshape(*Z)).flatten()
    if data == "L2":
        pf = lambda x, tau: prox.prox_ell2_denoising(x, tau, x0=b, x_scale=3.)
    else:
        pf = lambda x, tau: prox.prox_ell1_denoising(x, tau, x0=b, x_scale=3.)
    pg = prox.prox_huber_conj if reg == "Huber" else prox.prox_tv_conj
    return pd.PrimalDualSolver(prox_f=pf, prox_g_conj=pg, B=D, B_conj=Da,
                               L2=16, x0=b, x_scale=3.), b


def test_plan_recognises_native_wiring_through_lambdas():
    from nsol_amd import ops
    obs = np.arange(4 * 5 * 6, dtype=float).reshape(4, 5, 6)
    s, b = _wired_solver(obs, "L1", "Huber")
    plan = s.plan()
    assert plan is not None
    assert plan["shape"] == (4, 5, 6) and plan["dim"] == 3
    assert plan["w"] == (0.5, 0.5, 0.5)
    assert plan["flags"] == ops.PD_REG_HUBER | ops.PD_DATA_L1
    assert plan["data"] is b and plan["data_scale"] == 3.0
    s2, _ = _wired_solver(np.ones((7, 9)), "L2", "TV")
    assert s2.plan()["flags"] == 0 and s2.plan()["dim"] == 2


def test_plan_rejects_foreign_or_modified_callables():
    obs = np.ones((4, 5, 6))
    s, b = _wired_solver(obs)
    s._prox_f = lambda x, tau: np.asarray(x) * 2          # NumPy arithmetic
    assert s.plan() is None
    s, b = _wired_solver(obs)
    from nsol_amd.proximal_operators import ProximalOperators as prox
    s._prox_f = lambda x, tau: prox.prox_ell2_denoising(x, 2 * tau, x0=b)
    assert s.plan() is None                               # tau was modified
    s, b = _wired_solver(obs)
    B = s._B
    s._B = lambda x: B(B(x))                              # composition
    assert s.plan() is None
    s, b = _wired_solver(obs)
    s._B_conj = s._B                                      # not the adjoint
    assert s.plan() is None
    # a 3-D operator handed a flat array without the caller-side reshape
    import nsol_amd.linear_operators as LO
    grad3, _ = LO.LinearOperators3D().get_gradient_operators()
    s, b = _wired_solver(obs)
    s._B = grad3
    assert s.plan() is None


def test_solver_api_surface_without_gpu():
    obs = np.ones((3, 4, 8)) * 6.0
    s, b = _wired_solver(obs)
    assert s.get_alpha() == 0.01 and s.get_L2() == 16.0
    assert s.get_alg_type() == "ALG2" and s.get_iterations() == 10
    s.set_alpha(0.5), s.set_iterations(3), s.set_alg_type("ALG3")
    s.set_L2(12.)
    assert (s.get_alpha(), s.get_iterations(), s.get_alg_type(),
            s.get_L2()) == (0.5, 3, "ALG3", 12.)
    assert np.array_equal(s.get_x0(), b)          # x0 / x_scale * x_scale
    assert np.array_equal(s.get_x(), b)
    assert s.get_x_scale() == 3.0
    assert s.get_computational_time().total_seconds() == 0


def test_separable_detection_and_ndimage_centre():
    import nsol_amd.linear_operators as LO
    lo = LO.LinearOperators3D()
    A, A_adj = lo.get_gaussian_blurring_operators(np.diag([4., 4., 4.]))
    assert A is A_adj and A.separable
    assert [p[1].size for p in A._passes] == [13, 13, 13]
    assert [p[0] for p in A._passes] == [0, 1, 2]
    A2, _ = LO.LinearOperators2D().get_gaussian_blurring_operators(
        np.array([[2., .6], [.6, 1.]]))
    assert not A2.separable
    flipped, centre = LO._ndimage_convolve_params(np.arange(8.).reshape(2, 4))
    assert centre == [0, 1] and flipped[0, 0] == 7.0
    with pytest.raises(RuntimeError):
        LO.ConvolutionOperator(3, np.ones((3, 3)))


def test_synthetic_volume_matches_oracle_copy():
    from nsol_amd.synthetic import synth_volume
    from oracle import nsol_oracle as orc
    for kind in ("clean", "gauss", "sp"):
        assert np.array_equal(synth_volume(16, 2, kind),
                              orc.synth_volume(16, 2, kind))


@pytest.mark.parametrize("lo,hi", [(0.0, np.inf), (-np.inf, np.inf),
                                   (0.0, 1.5), (-np.inf, 0.7)])
@pytest.mark.parametrize("n,iters", [(5, 8), (40, 3), (40, 25), (300, 8),
                                     (300, 25)])
def test_lbfgsb_iteration_logic_tracks_scipy(lo, hi, n, iters):
    """nsol_amd.lbfgsb (the product's L-BFGS-B iteration logic), driven with a
    NumPy backend, reproduces scipy.optimize.minimize(method='L-BFGS-B'):
    same iterates, iteration count and function-evaluation count."""
    import scipy.optimize
    from nsol_amd import lbfgsb
    from lbfgsb_numpy_backend import NumpyBackend
    rng = np.random.default_rng(n + iters)
    A = rng.standard_normal((n + 5, n))
    b = 3.0 * rng.standard_normal(n + 5)
    c = rng.standard_normal(n)

    def fg(x):
        r = A @ x - b
        z = r * r
        return (float(np.sum(np.sqrt(1 + z) - 1)) +
                0.05 * float(np.sum((x - c) ** 4)),
                A.T @ (r / np.sqrt(1 + z)) + 0.2 * (x - c) ** 3)
    x0 = 2.0 * rng.standard_normal(n) + 1.0
    ref = scipy.optimize.minimize(
        fg, x0, jac=True, method="L-BFGS-B",
        bounds=scipy.optimize.Bounds(np.full(n, lo), np.full(n, hi)),
        options={"maxiter": iters})
    x, info = lbfgsb.minimize(fg, x0, lo, hi, NumpyBackend(), maxiter=iters)
    assert info["nit"] == ref.nit and info["nfev"] == ref.nfev
    assert np.linalg.norm(x - ref.x) <= 1e-10 * max(np.linalg.norm(ref.x), 1)


def test_lbfgsb_restarts_from_a_returned_point_without_reevaluating_it():
    """minimize(..., start=(f, g)): a second solve of the same objective from the
    point (and with the cost and gradient) the first one returned walks exactly the
    iterates of a second solve that evaluates its starting point again, with one
    evaluation fewer -- what ADMMLinearSolver does with minimizer='L-BFGS-B'."""
    from nsol_amd import lbfgsb
    from lbfgsb_numpy_backend import NumpyBackend
    rng = np.random.default_rng(3)
    n = 60
    A = rng.standard_normal((n + 5, n))
    b = 3.0 * rng.standard_normal(n + 5)
    calls = []

    def fg(x):
        calls.append(1)
        r = A @ x - b
        z = r * r
        return float(np.sum(np.sqrt(1 + z) - 1)), A.T @ (r / np.sqrt(1 + z))
    x0 = rng.standard_normal(n) + 1.0
    x1, info1 = lbfgsb.minimize(fg, x0, 0.0, np.inf, NumpyBackend(), maxiter=4)
    assert info1["fun"] == fg(x1)[0] and np.array_equal(info1["jac"], fg(x1)[1])
    del calls[:]
    xa, ia = lbfgsb.minimize(fg, x1, 0.0, np.inf, NumpyBackend(), maxiter=4)
    plain = len(calls)
    del calls[:]
    xb, ib = lbfgsb.minimize(fg, x1, 0.0, np.inf, NumpyBackend(), maxiter=4,
                             start=(info1["fun"], info1["jac"]))
    assert len(calls) == plain - 1
    assert np.array_equal(xa, xb) and ia["nit"] == ib["nit"] and ia["nfev"] == ib["nfev"]


def test_shard_indices():
    from nsol_amd.batch import shard_indices
    assert shard_indices(8, 1, 4) == [1, 5]
    assert shard_indices(8, 0, 1) == list(range(8))
    assert sorted(sum((shard_indices(7, r, 3) for r in range(3)), [])) == \
        list(range(7))


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from nsol_amd.batch import solve_batch
dist.init_process_group("gloo", rank=int(os.environ["RANK"]),
                        world_size=int(os.environ["WORLD_SIZE"]))
calls = []
def solve_one(i):
    calls.append(i)
    return torch.full((5,), float(i) * 10.0) + torch.arange(5.)
out = solve_batch(solve_one, 5)
rank = dist.get_rank()
assert calls == list(range(rank, 5, 2)), calls
if rank == 0:
    assert len(out) == 5
    for i, t in enumerate(out):
        assert torch.equal(t, torch.full((5,), float(i) * 10.0)
                           + torch.arange(5.)), (i, t)
    print("BATCH_OK")
else:
    assert out is None
dist.destroy_process_group()
'''


def test_solve_batch_gloo_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = 29500 + os.getpid() % 2000
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "BATCH_OK" in outs[0]


def test_solve_batch_with_fewer_volumes_than_ranks(tmp_path):
    """A rank that owns no volume takes part in the gather with a dummy of the
    agreed shape (it used to raise while the others waited in the collective)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER_FEW % ROOT)
    port = 31500 + os.getpid() % 2000
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "FEW_OK" in outs[0]


WORKER_FEW = '''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from nsol_amd.batch import solve_batch
dist.init_process_group("gloo", rank=int(os.environ["RANK"]),
                        world_size=int(os.environ["WORLD_SIZE"]))
out = solve_batch(lambda i: torch.full((7,), 3.0 + i), 1)
rank = dist.get_rank()
if rank == 0:
    assert len(out) == 1 and torch.equal(out[0], torch.full((7,), 3.0))
    print("FEW_OK")
else:
    assert out is None
assert solve_batch(lambda i: None, 0) == ([] if rank == 0 else None)
dist.destroy_process_group()
'''


def _bench(*argv, **env):
    e = dict(os.environ, **env)
    if "NSOL_KFD_TOPOLOGY" in env:      # the stated topology is all there is
        for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES",
                  "CUDA_VISIBLE_DEVICES"):
            e.pop(k, None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env:
            e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] +
                          list(argv), env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=300)


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` with no external launcher: two child ranks
    rendezvous (gloo), reduce, gather; rank 0 prints the one line.  --dry-run
    skips the GPU work so the plumbing is testable here."""
    import json
    r = _bench("--gpus", "2", "--dry-run", "--steps", "7", "--warmup", "1")
    assert r.returncode == 0, r.stderr.decode()
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["steps"] == 7


def test_bench_refuses_a_silent_single_gpu_run():
    # under a launcher whose world size disagrees with --gpus
    r = _bench("--gpus", "8", "--dry-run", WORLD_SIZE="1", RANK="0")
    assert r.returncode == 2 and b"WORLD_SIZE" in r.stderr
    # bare, RCCL backend, fewer devices than ranks (none in this container)
    import torch
    if torch.cuda.device_count() < 3:
        r = _bench("--gpus", "3")
        assert r.returncode == 2 and b"visible" in r.stderr
    # a batch that does not divide over the ranks
    r = _bench("--gpus", "2", "--batch", "3", WORLD_SIZE="2", RANK="0")
    assert r.returncode == 2 and b"multiple" in r.stderr


def _fake_kfd(tmp_path, simd_counts):
    for i, simd in enumerate(simd_counts):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(
            "cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n"
            % (0 if simd else 64, simd))
    return str(tmp_path)


def test_launcher_counts_gpus_without_the_hip_runtime(tmp_path, monkeypatch):
    """The parent of `--gpus N` reads the KFD topology (a node with SIMDs is a
    GPU) instead of asking torch / HIP, which would initialise the runtime in
    a process that then starts the ranks."""
    import bench
    root = _fake_kfd(tmp_path, [0, 0, 1024, 1024, 1024])   # 2 CPU sockets + 3 GPUs
    monkeypatch.setenv("NSOL_KFD_TOPOLOGY", root)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES",
                "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpus() == 2
    monkeypatch.setenv("NSOL_KFD_TOPOLOGY", str(tmp_path / "absent"))
    assert bench.visible_gpus() is None                     # unknown: ranks check
    src = open(os.path.join(ROOT, "bench.py")).read()
    launcher = src[src.index("def visible_gpus"):src.index("def parse_args")]
    assert "import torch" not in launcher and "device_count" not in launcher


def test_rccl_launcher_branch_against_a_stated_topology(tmp_path):
    """`--backend nccl` in the parent: refuses more ranks than the topology has
    GPUs before starting anything, starts them otherwise (dry run: the ranks
    themselves then rendezvous on gloo)."""
    import json
    root = _fake_kfd(tmp_path, [0, 1024, 1024])
    r = _bench("--gpus", "3", "--dry-run", "--backend", "nccl",
               NSOL_KFD_TOPOLOGY=root)
    assert r.returncode == 2 and b"only 2 GPU(s) visible" in r.stderr
    r = _bench("--gpus", "2", "--dry-run", "--backend", "nccl",
               NSOL_KFD_TOPOLOGY=root)
    assert r.returncode == 0, r.stderr.decode()
    d = json.loads([ln for ln in r.stdout.decode().splitlines()
                    if ln.startswith("{")][0])
    assert d["ranks_joined"] == 2


def test_minres_scalars_restate_scipys_lsmr_stopping_tests():
    """nsol_amd/lsmr.py runs LSMR as Lanczos / MINRES on the normal equations;
    SciPy's stopping quantities (lsmr.py:416-449 behind
    tikhonov_linear_solver.py:146-158) are rebuilt from the MINRES scalars
    (MinresCoefficients.lsmr_tests).  Host arithmetic only: Lanczos in NumPy on
    small dense systems, every iterate against the oracle's LSMR, the tests
    against their definitions, and the iteration at which a 7-dimensional Krylov
    space is exhausted (SciPy: istop 5 at itn 7) found at the same step."""
    import math
    from nsol_amd.lsmr import MinresCoefficients
    from oracle import nsol_oracle as orc
    for seed in range(6):
        rng = np.random.default_rng(seed)
        m, n = 30, 7
        Ab, b = rng.standard_normal((m, n)), rng.standard_normal(m)
        M, g = Ab.T @ Ab, Ab.T @ b
        beta1 = np.linalg.norm(g)
        V, betas, vprev = [g / beta1], [beta1], np.zeros(n)
        co = MinresCoefficients(12, beta1)
        fired = None
        for k in range(1, 10):
            v = V[-1]
            y = M @ v
            alfa = v @ y
            y = y - alfa * v - betas[-1] * vprev * (k > 1)
            beta = np.linalg.norm(y)
            co.step(alfa, beta)
            x = np.stack(V, 1) @ co.x[:k]
            xo, istop_o, itn_o = orc.lsmr(lambda u: Ab @ u, lambda u: Ab.T @ u,
                                          b, n, k)
            test1, test2, t1 = co.lsmr_tests(float(b @ b))
            r = b - Ab @ x
            if itn_o == k:
                assert np.linalg.norm(x - xo) <= 1e-13 * np.linalg.norm(xo)
            assert abs(test1 - np.linalg.norm(r) / np.linalg.norm(b)) < 1e-13
            if k < n:
                true2 = np.linalg.norm(Ab.T @ r) / np.linalg.norm(r)
                assert abs(test2 * math.sqrt(sum(co.alfas)) - true2) < 1e-10 * true2
                assert istop_o == 7 and not (1 + test2 <= 1) and not (1 + t1 <= 1)
            elif fired is None and 1 + test2 <= 1:
                fired = k
                assert (istop_o, itn_o) == (5, k)
            vprev = v
            V.append(y / beta)
            betas.append(beta)
        assert fired == n


def test_bridge_tells_gpu_failures_from_numpy_only_callables():
    """BridgedCallable falls back to the host round trip only for callables
    that cannot take a tensor; a failing launch or an out-of-memory error is
    re-raised (nsol_amd/bridge.py)."""
    from nsol_amd.bridge import _is_gpu_failure
    from nsol_amd._lib import NsolHipError
    assert _is_gpu_failure(NsolHipError("nsol_grad failed with hipError_t 719"))
    assert _is_gpu_failure(RuntimeError("HIP out of memory. Tried to allocate"))
    assert _is_gpu_failure(RuntimeError("HIP error: an illegal memory access"))
    assert not _is_gpu_failure(TypeError("can't convert cuda:0 device type "
                                         "tensor to numpy"))
    assert not _is_gpu_failure(AttributeError("'Tensor' has no attribute x"))
    assert not _is_gpu_failure(ValueError("operands could not be broadcast"))


def test_data_term_cache_sees_in_place_changes():
    from nsol_amd.proximal_operators import _fingerprint
    a = np.arange(100000, dtype=np.float64)
    f0 = _fingerprint(a)
    a *= 2.0
    assert _fingerprint(a) != f0
    b = np.arange(10.0)
    f1 = _fingerprint(b)
    b[3] = -1.0
    assert _fingerprint(b) != f1


def test_persistent_kernel_tiling_rules():
    """nsol_pd_persist_ws_bytes (host logic only): the persistent kernel applies
    to volumes one tile of <= 1024 lanes per CU can cover, with rows of whole
    16-byte vectors; the workspace grows with the iteration count (the step
    sizes live in it)."""
    from nsol_amd import _lib, ops
    lib = _lib.load()
    ws = lib.nsol_pd_persist_ws_bytes
    assert ws(4, 3, 64, 64, 64, 200) > 0 and ws(8, 3, 64, 64, 64, 200) > 0
    assert ws(4, 2, 1, 256, 256, 50) > 0 and ws(4, 1, 1, 1, 4096, 10) > 0
    assert ws(4, 3, 64, 64, 64, 400) > ws(4, 3, 64, 64, 64, 200)
    assert ws(4, 3, 7, 10, 13, 10) == -1            # ragged rows
    assert ws(4, 3, 512, 512, 512, 10) == -1        # more tiles than CUs
    assert ws(4, 2, 64, 64, 64, 10) == -1           # ndim / extents mismatch
    assert ws(2, 3, 64, 64, 64, 10) == -1 and ws(4, 3, 64, 64, 64, 0) == -1
    # where one launch per run is expected to pay (tools/bench_persist.py)
    assert ops.persist_pays((64, 64, 64), 200) and ops.persist_pays((32, 32, 32), 16)
    assert not ops.persist_pays((256, 256), 50)     # 3.4 us launches win
    assert ops.persist_pays((1024, 1024), 100)
    assert not ops.persist_pays((64, 64, 64), 8)    # too few iterations
    assert not ops.persist_pays((128, 128, 128), 100)   # K = 3 kernel's range


def test_lsmr_solution_coefficients_follow_the_vector_recurrences():
    """lsmr_fused assembles x = sum_k a_k v_k once at the end instead of carrying
    h, hbar and x through every iteration (SciPy lsmr.py:352-364): the coefficient
    recurrences against the vector recurrences on random data, with a breakdown
    step (no new vector) in between."""
    import sys
    sys.modules.setdefault("torch", __import__("torch"))
    from nsol_amd.lsmr import SolutionCoefficients
    rng = np.random.default_rng(4)
    n, K = 50, 9
    V = rng.standard_normal((K + 1, n))
    coef = SolutionCoefficients(K + 1)
    h, hbar, x = V[0].copy(), np.zeros(n), np.zeros(n)
    newest = 0
    for k in range(K):
        c_hbar, c_x, c_h = rng.standard_normal(3)
        if k != 4:
            newest += 1                       # (k == 4: breakdown, v stays)
        hbar = h + c_hbar * hbar
        x = x + c_x * hbar
        h = V[newest] + c_h * h
        coef.step(c_hbar, c_x, c_h, newest)
        assert np.allclose(coef.x @ V, x, rtol=1e-12, atol=1e-12)
        assert np.allclose(coef.h @ V, h, rtol=1e-12, atol=1e-12)
        assert np.allclose(coef.hbar @ V, hbar, rtol=1e-12, atol=1e-12)
    Q, _ = np.linalg.qr(V.T)                  # orthonormal columns: ||x||^2 = sum a^2
    assert np.isclose(coef.normx2(), np.sum((Q @ coef.x) ** 2))


def test_minres_coefficients_reproduce_lsmr_iterates():
    """lsmr_normal runs LSMR as Lanczos on M = A'A + rho B'B with the Paige-Saunders
    rotations on the host (MinresCoefficients) and x assembled from the Lanczos
    vectors at the end: in exact arithmetic these are the iterates of SciPy's lsmr on
    [A; sqrt(rho) B] (tikhonov_linear_solver.py:146-158 on :226-274), iteration by
    iteration -- checked here in float64 NumPy, where the two agree to rounding."""
    import scipy.sparse.linalg as sla
    from nsol_amd.lsmr import MinresCoefficients
    rng = np.random.default_rng(11)
    m, n, K = 70, 30, 10
    A = rng.standard_normal((m, n)) / np.sqrt(m)
    B = np.eye(n) - np.eye(n, k=1)                      # a difference operator
    for rho in (0.5, 0.1):
        Ahat = np.vstack([A, np.sqrt(rho) * B])
        bhat = np.concatenate([rng.standard_normal(m), np.sqrt(rho) * rng.standard_normal(n)])
        M = Ahat.T @ Ahat
        g = Ahat.T @ bhat
        beta = float(np.linalg.norm(g))
        V = [g / beta]
        co = MinresCoefficients(K + 1, beta)
        for j in range(K):
            w = M @ V[j] - (beta * V[j - 1] if j > 0 else 0.0)
            alfa = float(V[j] @ w)
            w = w - alfa * V[j]
            beta = float(np.linalg.norm(w))
            co.step(alfa, beta)
            V.append(w / beta)
            x = co.x[:j + 1] @ np.array(V[:j + 1])
            ref = sla.lsmr(Ahat, bhat, atol=0.0, btol=0.0, conlim=0.0, maxiter=j + 1)[0]
            assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref), (rho, j)
        assert co.itn == K and co.gmax / co.gmin < 1e3


def test_data_cache_serves_only_what_is_provably_unchanged():
    """nsol_amd/_caches.py: an entry is tied to the storage it was derived from (weak
    reference: dropped when the memory is freed, never keeping it alive) and to torch's
    version counter, which ops._wrote bumps for every kernel write; anything else is
    the caller's to announce with nsol_amd.invalidate_caches()."""
    import gc
    import torch
    import nsol_amd
    from nsol_amd import _caches, ops
    c = _caches.DataCache(2)
    b = torch.arange(8, dtype=torch.float32)
    assert c.lookup((b,), 2.0) is None
    val = c.store((b,), 2.0, "b/2")
    assert c.lookup((b,), 2.0) is val
    assert c.lookup((b.view(2, 4).view(-1),), 2.0) is val   # a view of the same memory
    assert c.lookup((b,), 3.0) is None                      # another x_scale
    assert c.lookup((b[1:],), 2.0) is None                  # another window
    b.add_(1.0)                                             # torch sees this write
    assert c.lookup((b,), 2.0) is None
    c.store((b,), 2.0, "new")
    ops._wrote(b)                                           # a kernel wrote through data_ptr()
    assert c.lookup((b,), 2.0) is None
    c.store((b,), 2.0, "newer")
    assert c.lookup((b,), 2.0) == "newer"
    nsol_amd.invalidate_caches()
    assert c.entries == [] and c.lookup((b,), 2.0) is None
    # freed memory takes its entries along, and the cache never holds it
    import weakref
    c.store((b,), 2.0, "x")
    alive = weakref.ref(b.untyped_storage())
    del b
    gc.collect()
    assert alive() is None and c.entries == []
    # only the newest `keep` entries stay
    ts = [torch.zeros(4) for _ in range(3)]
    for i, t in enumerate(ts):
        c.store((t,), None, i)
    assert len(c.entries) == 2 and c.lookup((ts[0],)) is None and c.lookup((ts[2],)) == 2


def test_a_residual_lost_to_cancellation_is_unknown_not_zero():
    """MinresCoefficients.lsmr_tests forms |r_k|^2 = |b|^2 - 2 beta_1 z_1 + z'T z by
    cancellation.  On a consistent system with scalars summed over float32 vectors the
    true value (<= 1e-9 |b|^2 here) is below the rounding of its terms and may come
    out <= 0: SciPy (lsmr.py:416-449, atol = btol = 0) keeps a positive estimate and
    iterates on, so the restated tests must not report istop 1 there."""
    from nsol_amd.lsmr import MinresCoefficients, _scipy_stop
    rng = np.random.default_rng(2)
    n = 7
    Ab = (np.eye(n) + 0.1 * rng.standard_normal((n, n))).astype(np.float32)
    b = (Ab @ rng.standard_normal(n).astype(np.float32)).astype(np.float32)
    M, g = (Ab.T @ Ab).astype(np.float32), (Ab.T @ b).astype(np.float32)
    beta1 = float(np.linalg.norm(g))
    v, vprev, beta = g / np.float32(beta1), np.zeros(n, np.float32), beta1
    co = MinresCoefficients(12, beta1)
    eps32 = float(np.finfo(np.float32).eps)
    normb2 = float(b @ b)
    seen_unknown = False
    for k in range(1, n + 1):
        y = (M @ v).astype(np.float32)
        alfa = float(v @ y)
        y = (y - np.float32(alfa) * v - np.float32(beta) * vprev * (k > 1)).astype(np.float32)
        beta_new = float(np.linalg.norm(y))
        co.step(alfa, beta_new)
        t32 = co.lsmr_tests(normb2, eps32)
        if np.isnan(t32[0]):
            seen_unknown = True
            assert _scipy_stop(co, normb2, eps32) == 0          # iterate on, as SciPy does
            # (what the same scalars say without the guard is not to be trusted)
            assert co.lsmr_tests(normb2)[0] < 1e-2
        vprev, v, beta = v, y / np.float32(beta_new), beta_new
    assert seen_unknown


def test_iterate_20_of_the_edge_case_depends_on_rounding_at_the_1e10_level():
    """Where the float64 gate of test_normal_equations_lsmr_at_the_edge_of_its_guard
    (tests/test_gpu_parity.py: 1e-8, not 1e-12) comes from, derived here on the CPU.

    The case: sigma = 2 blur at 32^3, B = identity, weight on the guard's bound
    (tests/golden/cfg4.npz: y_32, ratio_32, the reference's tk_ident_edge_{10,20,32};
    tikhonov_linear_solver.py:146-158 with SciPy's LSMR).  The same iterates from
      (a) Lanczos / MINRES on the normal equations (nsol_amd/lsmr.py's recurrence,
          MinresCoefficients) with A evaluated by FFT,
      (b) the same with A evaluated separably along x, y, z (what the HIP blur does),
      (c) (a) with every Lanczos vector reorthogonalised twice against all before it,
      (d) scipy.sparse.linalg.lsmr on the augmented operator, A by FFT,
      (e) the reference's own output (dense scipy.ndimage taps; the golden).
    The three evaluations of A agree to 2e-15.  At k = 10 and k = 32 all five iterates
    agree to 1e-13; at k = 20 the Krylov process has just found the dominant
    eigenvalues, its vectors have lost orthogonality (max |v_i'v_j| > 1e-2), and the
    iterate moves by 1e-10 ... 1e-9 with the rounding of the operator and the form of
    the recurrence -- five orders of magnitude more than at k = 10 and 32.  The MI355X
    run lands 1.1e-9 from the golden there (profiles/*_parity_errors.json): a gate of
    1e-8 holds the implementation to one decade above that sensitivity, 1e-9 sat
    inside it."""
    import math
    import warnings
    import scipy.sparse.linalg as spla
    import nsol_amd.kernels as Kn
    from nsol_amd.lsmr import MinresCoefficients
    from oracle import nsol_oracle as orc
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "cfg4.npz")))
    n = 32
    shape, N = (n, n, n), n ** 3
    y = g["y_32"].astype(np.float64).reshape(-1)
    xs = float(y.max())
    b = y / xs
    alpha = 0.1 * float(g["ratio_32"])
    sa = math.sqrt(alpha)
    taps = orc.gaussian_taps(3, np.diag([4.0, 4.0, 4.0]))            # dense 13^3
    c = taps.shape[0] // 2
    K = np.zeros(shape)
    idx = (np.arange(taps.shape[0]) - c) % n
    K[np.ix_(idx, idx, idx)] = taps
    FK = np.fft.rfftn(K)
    ax3 = (0, 1, 2)
    A_fft = lambda v: np.fft.irfftn(np.fft.rfftn(v.reshape(shape)) * np.conj(FK),
                                    s=shape, axes=ax3).reshape(-1)
    t1 = np.asarray(Kn.Kernels1D().get_gaussian(4.0), dtype=np.float64).reshape(-1)

    def A_sep(v):                       # periodic, symmetric taps: x, then y, then z
        v = v.reshape(shape)
        for ax in (2, 1, 0):
            acc = np.zeros_like(v)
            for i, w in enumerate(t1):
                acc = acc + w * np.roll(v, -(i - t1.size // 2), axis=ax)
            v = acc
        return v.reshape(-1)
    probe = np.random.default_rng(0).standard_normal(N)
    _, _, A_ref, _ = orc.flat_operators(shape, None, np.diag([4.0, 4.0, 4.0]))
    for A in (A_fft, A_sep):
        assert np.linalg.norm(A(probe) - A_ref(probe)) <= 1e-14 * np.linalg.norm(probe)

    def lanczos(A, kmax, reorth):
        gv = A(b)                                       # (A is symmetric: A' = A)
        beta1 = np.linalg.norm(gv)
        V, co, beta, vprev = [gv / beta1], MinresCoefficients(kmax + 1, beta1), beta1, 0.0
        out = {}
        for k in range(1, kmax + 1):
            v = V[-1]
            w = A(A(v)) + alpha * v
            alfa = float(v @ w)
            w = w - alfa * v - beta * vprev
            if reorth:
                for _ in range(2):
                    for u in V:
                        w = w - (u @ w) * u
            beta = float(np.linalg.norm(w))
            co.step(alfa, beta)
            out[k] = np.stack(V, 1) @ co.x[:k]
            vprev = v
            V.append(w / beta)
        G = np.stack(V[:kmax], 1)
        return out, np.abs(G.T @ G - np.eye(kmax))
    xa, Ga = lanczos(A_fft, 32, False)
    xb, _ = lanczos(A_sep, 32, False)
    xc, Gc = lanczos(A_fft, 32, True)
    assert Gc.max() < 1e-12                              # (c) stays orthonormal
    assert Ga[:10, :10].max() < 1e-9 and Ga[:20, :20].max() > 1e-2
    op = spla.LinearOperator(
        (2 * N, N), dtype=np.float64,
        matvec=lambda v: np.concatenate([A_fft(v), sa * v]),
        rmatvec=lambda u: A_fft(u[:N]) + sa * u[N:])
    rhs = np.concatenate([b, np.zeros(N)])
    spread = {}
    for k in (10, 20, 32):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            xd = spla.lsmr(op, rhs, atol=0, btol=0, conlim=1e8, maxiter=k)[0]
        xe = g["tk_ident_edge_%d" % k].reshape(-1) / xs
        its = [np.clip(v, 0, np.inf) for v in (xa[k], xb[k], xc[k], xd)] + [xe]
        nrm = np.linalg.norm(xe)
        spread[k] = max(np.linalg.norm(p - q) / nrm for i, p in enumerate(its)
                        for q in its[i + 1:])
    assert spread[10] < 1e-13 and spread[32] < 1e-13
    assert 1e-10 < spread[20] < 1e-9
    assert spread[20] > 1e4 * max(spread[10], spread[32])
