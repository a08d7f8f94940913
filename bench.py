#!/usr/bin/env python3
"""Headline benchmark: Chambolle-Pock primal-dual iterations/sec on a synthetic
512^3 float32 TV-L2 denoising problem (BASELINE.json configs[2]), one volume
per GPU.

    python bench.py --gpus N --steps K --warmup W

A step is ONE primal-dual iteration over the whole volume (inputs already
resident in HBM); three consecutive iterations share one pass over memory
(k_pd_fusedk, temporal blocking of depth 3).

N > 1: one process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in
the environment) this process is one of the ranks; started bare, `--gpus N`
starts the N ranks itself (child processes, before anything here touches a GPU)
and fails if fewer than N devices are visible or a rank does not join.  Each
rank owns its own volume (weak scaling, no per-iteration collective); after the
timed region the results are gathered once on rank 0 over RCCL (gather_ms).

`--batch B` is BASELINE config 5 instead: B independent TV-L1 volumes (salt and
pepper) sharded over the ranks, each rank solving B / N of them one after the
other, ONE gather of all reconstructions at the end; the clock covers solves
and gather (strong scaling).

The timed region of K steps is repeated `--reps` times (default 5, each
bracketed by barrier + synchronize, max over ranks); the line reports the
median.  Rank 0 prints one JSON line.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_VOXEL = 44        # 11 float32 words: read xbar,x,b,p[3]; write p[3],x,xbar
PMC_TABLE = os.path.join("profiles", "pmc_by_config.json")


class HipEvents(object):
    """hipEvent timing on the stream the kernels are launched on."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [
            ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]

    def create(self):
        ev = ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(ev)) == 0
        return ev

    def record(self, ev, stream):
        assert self.hip.hipEventRecord(ev, ctypes.c_void_p(stream)) == 0

    def elapsed_ms(self, a, b):
        assert self.hip.hipEventSynchronize(b) == 0
        ms = ctypes.c_float()
        assert self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
        return float(ms.value)


def profiled_traffic(n, plan=None):
    """HBM bytes per launch of the headline kernel from the committed rocprofv3
    PMC passes (profiles/pmc_by_config.json, written by
    tools/summarize_profiles.py).  PMC counters cannot be read from inside this
    process: the figure belongs to ANOTHER run of the same binary, pinned to one
    kernel configuration, and is looked up by the configuration this run's
    tuner settled on.  Returns (bytes or None, source string or None)."""
    if not plan:
        return None, None
    try:
        table = json.load(open(os.path.join(ROOT, PMC_TABLE)))
        key = "k_pd_fusedk:%d:%d:%d:%d" % ((n,) + tuple(plan[:3]))
        rec = table.get(key)
        if rec:
            return float(rec["traffic_bytes_per_launch"]), \
                "from_profile: %s[%s] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE " \
                "passes of another run, same kernel configuration)" % (
                    PMC_TABLE, key)
    except Exception:
        pass
    return None, None


def cpu_baseline(sample_n, iters):
    """Reference-style NumPy/SciPy iteration (oracle port) on the host."""
    from oracle import nsol_oracle as orc
    vol = orc.synth_volume(sample_n, 0, "gauss")
    b = vol.reshape(-1)
    xs = float(vol.max())
    shape = vol.shape
    orc.pd_tvl2_refstyle(b, shape, 0.03, 1, 16.0, xs)          # warm-up
    t0 = time.time()
    orc.pd_tvl2_refstyle(b, shape, 0.03, iters, 16.0, xs)
    dt = time.time() - t0
    return iters / dt


def cpu_baseline_c(sample_n, iters):
    """The same loop as compiled C with OpenMP (oracle/pd_oracle.c, bit-identical
    to the NumPy oracle) on all host cores: what a tuned CPU port would do."""
    from oracle import nsol_oracle as orc, c_oracle
    vol = orc.synth_volume(sample_n, 0, "gauss")
    b = vol.reshape(-1)
    c_oracle.primal_dual_denoise(b, vol.shape, "TV", "L2", 0.03, 1, 16.0)
    t0 = time.time()
    c_oracle.primal_dual_denoise(b, vol.shape, "TV", "L2", 0.03, iters, 16.0)
    dt = time.time() - t0
    return iters / dt, c_oracle.threads()


def config4_line(n, runs=3, cpu_sample=64):
    """BASELINE config 4 (sigma = 2 deconvolution, ADMMLinearSolver alpha = 0.01,
    rho = 0.1, 10 ADMM x 10 inner iterations, float32) measured in THIS process so that
    the driver's line carries it: both branches (LSMR / linear loss; L-BFGS-B / Huber
    loss), each with seconds per run (median of `runs` runs after one that carries the
    one-time set-up), its `roofline` (the dominant kernel's duration INSIDE a run:
    event pairs around every C-ABI entry, bench_admm.kernels_in_run) and its
    `cpu_baseline` (one ADMM iteration of the reference-style CPU path at cpu_sample^3,
    extrapolated per voxel).  bench_admm.py prints the same record on its own."""
    import bench_admm
    rec = bench_admm.measure(n, 10, 10, "lsmr", "linear", runs + 1, cpu_sample)
    hub = bench_admm.measure(n, 10, 10, "L-BFGS-B", "huber", runs + 1, cpu_sample)

    def brief(r):
        # (the per-entry table stays, without its min / max columns)
        roof = dict(r["roofline"])
        roof["kernels"] = {k: {f: v[f] for f in ("kernel", "launches_per_run",
                                                 "avg_launch_ms", "ms_per_run",
                                                 "bytes_per_voxel", "frac")
                               if f in v}
                           for k, v in roof["kernels"].items()}
        return {"metric": r["metric"], "value": r["value"], "unit": r["unit"],
                "seconds_per_run": r["seconds_per_run"], "runs_s": r["runs"][1:],
                "first_run_s": r["runs"][0], "workload": r["config"]["workload"],
                "minimizer": r["config"]["minimizer"],
                "data_loss": r["config"]["data_loss"],
                "execution": r["config"]["execution"],
                "lsmr_step": r["config"]["lsmr_form"],
                "inner_solves": r["config"]["inner_solves"],
                "result_finite": r["finite"], "roofline": roof,
                "cpu_baseline": r.get("cpu_baseline")}
    out = brief(rec)
    out["huber_branch"] = brief(hub)
    return out


# ------------------------------------------------------------------ launcher
def visible_gpus():
    """GPUs of this node WITHOUT loading the HIP runtime: the launcher goes on
    to start the ranks as child processes, and a parent that has initialised a
    GPU must not do that.  Counted from the KFD topology in sysfs (a node with
    SIMDs is a GPU), capped by a *_VISIBLE_DEVICES list.  None = unknown (no
    sysfs here): every rank repeats the check with the runtime's own count and
    refuses a world larger than it."""
    root = os.environ.get("NSOL_KFD_TOPOLOGY",
                          "/sys/class/kfd/kfd/topology/nodes")
    try:
        nodes = os.listdir(root)
    except OSError:
        return None
    n = 0
    for d in nodes:
        try:
            with open(os.path.join(root, d, "properties")) as f:
                props = f.read()
        except OSError:
            continue
        for line in props.splitlines():
            key, _, val = line.partition(" ")
            if key == "simd_count" and val.strip().isdigit() and int(val) > 0:
                n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES",
                "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var) is not None:
            n = min(n, len([t for t in os.environ[var].split(",")
                            if t.strip()]))
    return n


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without an external launcher: start N ranks as
    child processes (this parent never initialises a GPU), pass rank 0's JSON
    line through, fail if any rank fails."""
    n = args.gpus
    # (--dry-run takes this branch only against a stated topology: the CPU test
    # of the RCCL launcher path; a bare dry run rehearses on gloo with no GPU)
    if args.backend == "nccl" and (not args.dry_run or
                                   "NSOL_KFD_TOPOLOGY" in os.environ):
        have = visible_gpus()
        if have is not None and have < n:
            sys.stderr.write(
                "bench.py: --gpus %d but only %d GPU(s) visible; refusing to "
                "report a %d-rank number from fewer devices (use --backend "
                "gloo to rehearse)\n" % (n, have, n))
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + argv, env=env,
            stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + args.launch_timeout
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
        if rc != 0 or time.time() > deadline:
            for p in pending:           # a rank died or the job hangs: exact PIDs
                p.kill()
            for p in pending:
                p.wait()
            if rc == 0:
                rc = 3
                sys.stderr.write("bench.py: ranks did not finish within %d s\n"
                                 % args.launch_timeout)
            break
        time.sleep(0.05)
    if rc != 0:
        sys.stderr.write("bench.py: %d-rank run failed (exit %d)\n" % (n, rc))
    return rc


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--copy-back", action="store_true",
                    help="copy the final iterate back into x after an odd number "
                         "of multi-iteration launches instead of letting x and "
                         "its scratch volume trade storage (A/B runs)")
    ap.add_argument("--reps", type=int, default=5,
                    help="repetitions of the timed K-step region (median is "
                         "reported)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--data", default=None, choices=["L2", "L1"])
    ap.add_argument("--batch", type=int, default=0,
                    help="BASELINE config 5: this many independent TV-L1 "
                         "volumes sharded over the ranks, one gather at the "
                         "end inside the clock (0 = one volume per rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config4", action="store_true",
                    help="skip the extra key with BASELINE config 4's time "
                         "(three 10 x 10 ADMM runs, about 2 s)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the replay of the timed schedule through the "
                         "one-iteration kernel (profiling passes: keeps the "
                         "kernel trace to the timed launches)")
    ap.add_argument("--pdk", default="",
                    help="waves:ntx:zchunk -- pin the depth-3 kernel's "
                         "configuration instead of tuning (profiling passes)")
    ap.add_argument("--verbose-tuning", action="store_true",
                    help="print the online tuner's decision (stderr)")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="edge length of the volume the CPU baseline is timed "
                         "on (0 = the benchmark's own size: about 20 s of CPU "
                         "work at 512^3)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo only to rehearse N > 1 on a box with fewer "
                         "GPUs than ranks (collectives then go through host "
                         "memory)")
    ap.add_argument("--launch-timeout", type=int, default=1500)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / collectives only, no GPU "
                         "work (CPU test of the N > 1 plumbing)")
    return ap.parse_args(argv)


def median(v):
    v = sorted(v)
    m = len(v) // 2
    return v[m] if len(v) % 2 else 0.5 * (v[m - 1] + v[m])


def dry_run(args, world, rank):
    """The N > 1 plumbing without a GPU: rendezvous, barrier, MAX-reduce of a
    time, one gather; rank 0 prints a line."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    joined = torch.ones(1, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(joined)
        res = torch.full((4,), float(rank))
        bucket = [torch.empty_like(res) for _ in range(world)] \
            if rank == 0 else None
        dist.gather(res, gather_list=bucket, dst=0)
        if rank == 0:
            assert [int(b[0]) for b in bucket] == list(range(world))
        every = [torch.empty_like(t) for _ in range(world)]     # per-rank ms
        dist.all_gather(every, torch.tensor([float(rank)], dtype=torch.float64))
        assert [int(e.item()) for e in every] == list(range(world))
    if rank == 0:
        assert int(joined.item()) == world and float(t.item()) == world
        print(json.dumps({"dry_run": True, "n_gpus": world,
                          "ranks_joined": int(joined.item()),
                          "steps": args.steps, "warmup": args.warmup}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args, argv)
    world = int(env_world or "1")
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch with "
                         "--nproc-per-node equal to --gpus\n"
                         % (args.gpus, world))
        return 2
    rank = int(os.environ.get("RANK", "0"))
    if args.dry_run:
        return dry_run(args, world, rank)
    if args.batch and args.batch % world:
        sys.stderr.write("bench.py: --batch %d is not a multiple of %d ranks\n"
                         % (args.batch, world))
        return 2

    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    if ndev < 1 or (args.backend == "nccl" and ndev < world):
        sys.stderr.write("bench.py: %d rank(s) on %d visible GPU(s)\n"
                         % (world, ndev))
        return 2
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % ndev
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(
                "cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    on_host = world > 1 and args.backend == "gloo"

    from nsol_amd import ops, _lib
    from nsol_amd.primal_dual_solver import step_schedule
    from nsol_amd.synthetic import synth_volume
    _lib.load()
    if args.verbose_tuning:
        _lib.set_param("pdk_verbose", 2)
    pinned = None
    if args.pdk:
        pinned = tuple(int(t) for t in args.pdk.split(":"))
        _lib.set_param("pdk_nw", pinned[0])
        _lib.set_param("pdk_ntx", pinned[1])
        _lib.set_param("pdk_zchunk", pinned[2])

    data = args.data or ("L1" if args.batch else "L2")
    n = args.size
    shape = (n, n, n)
    nvox = n ** 3
    alpha = 0.03 if data == "L2" else 0.6
    kind = "gauss" if data == "L2" else "sp"
    dev = torch.device("cuda", local_rank)
    # volume i of the batch -> rank i % world (nsol_amd.batch.shard_indices);
    # without --batch every rank owns exactly volume `rank`
    mine = list(range(rank, args.batch, world)) if args.batch else [rank]
    inputs, scales = [], []
    for i in mine:                      # made on the host, resident in HBM
        vol = synth_volume(n, seed=i, kind=kind, dtype=np.float32)
        scales.append(float(vol.max()))
        b = torch.from_numpy(vol.reshape(-1)).to(dev)
        del vol
        inputs.append(ops.scale(b, scales[-1], divide=True))   # b~ = b / x_scale
        del b
    bt = inputs[0]
    x = bt.clone()
    x_alt = torch.empty_like(bt)      # x ping-pong of the multi-iteration kernels
    xbar = [bt.clone(), torch.empty_like(bt)]
    p = [torch.zeros(3 * nvox, dtype=torch.float32, device=dev)
         for _ in range(2)]
    w = (1.0, 1.0, 1.0)
    lmbda = 1.0 / alpha
    reps = max(1, args.reps)
    flags = ops.PD_REG_TV | (ops.PD_DATA_L1 if data == "L1"
                             else ops.PD_DATA_L2)
    state = {"slot": 0}

    def run(bt, sig, ta, th, first, count, p_is_zero):
        a = state["slot"]                 # slot holding the current xbar / p
        end = ops.pd_run(xbar[a], xbar[1 - a], x, bt, p[a], p[1 - a], shape, w,
                         lmbda, sig[first:first + count],
                         ta[first:first + count], th[first:first + count],
                         p_is_zero, 0.05, flags, x_alt=x_alt,
                         swap_ok=not args.copy_back)
        state["slot"] = a ^ end

    def reset(bt):
        x.copy_(bt)
        xbar[0].copy_(bt)
        state["slot"] = 0

    # Plan pass (untimed, like creating an FFT plan): the depth-3 kernel tunes
    # its footprint shape online during the first few dozen launches on a new
    # problem shape.  Run until it has settled, then reset the state.
    psig, pta, pth = step_schedule("ALG2", 16.0, lmbda, 60)   # its own schedule:
    for _ in range(8):                  # independent of --steps / --warmup
        ops.pd_run(xbar[0], xbar[1], x, bt, p[0], p[1], shape, w, lmbda, psig,
                   pta, pth, True, 0.05, flags, x_alt=x_alt)
        torch.cuda.synchronize()
        if pinned or ops.pd_fusedk_tuned(x, shape) != 0:
            break
    plan = pinned or ops.pd_fusedk_plan(x, shape)
    reset(bt)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64,
                         device="cpu" if on_host else dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_results(res):
        """The one collective: every rank's reconstructions to rank 0."""
        if on_host:
            res = res.cpu()
        bucket = [torch.empty_like(res) for _ in range(world)] \
            if rank == 0 else None
        dist.gather(res, gather_list=bucket, dst=0)
        torch.cuda.synchronize()
        return bucket

    ev = HipEvents()
    stream = torch.cuda.current_stream().cuda_stream
    triples = args.steps // 3
    rep_s, rep_kernel_ms = [], []
    gather_ms = None

    if not args.batch:
        total = args.warmup + reps * args.steps
        sig, ta, th = step_schedule("ALG2", 16.0, lmbda, total)
        if args.warmup > 0:
            run(bt, sig, ta, th, 0, args.warmup, True)
        first = args.warmup
        for r in range(reps):
            # One repetition is ONE enqueue of args.steps iterations; an event
            # after the last full group of three separates the launches of the
            # dominant kernel (k_pd_fusedk, three iterations per launch) from
            # the one or two trailing iterations (k_pd_fused2 / k_pd_fused).
            e0, em, e1 = ev.create(), ev.create(), ev.create()
            fence()
            t0 = time.perf_counter()
            ev.record(e0, stream)
            if triples:
                run(bt, sig, ta, th, first, 3 * triples, first == 0)
            ev.record(em, stream)
            if args.steps > 3 * triples:
                run(bt, sig, ta, th, first + 3 * triples,
                    args.steps - 3 * triples, first == 0 and triples == 0)
            ev.record(e1, stream)
            fence()
            rep_s.append(max_over_ranks(time.perf_counter() - t0))
            rep_kernel_ms.append(ev.elapsed_ms(e0, em) / triples if triples
                                 else ev.elapsed_ms(e0, e1))
            first += args.steps
        if world > 1:                                 # the one collective
            fence()
            g0 = time.perf_counter()
            bucket = gather_results(ops.scale(x, scales[0]))
            gather_ms = (time.perf_counter() - g0) * 1e3
            del bucket
    else:
        # config 5: each rank solves its share of the batch, volume after
        # volume (args.steps iterations each), then ONE gather; the clock
        # covers all of it
        sig, ta, th = step_schedule("ALG2", 16.0, lmbda, args.steps)
        results = torch.empty(len(mine) * nvox, dtype=torch.float32, device=dev)
        if args.warmup > 0:
            run(bt, sig, ta, th, 0, min(args.warmup, args.steps), True)
        gathers = []
        for r in range(reps):
            e0, em = ev.create(), ev.create()
            fence()
            t0 = time.perf_counter()
            kms = 0.0
            for j, bj in enumerate(inputs):
                reset(bj)
                if j == 0:
                    ev.record(e0, stream)
                if triples:
                    run(bj, sig, ta, th, 0, 3 * triples, True)
                if j == 0:
                    ev.record(em, stream)
                if args.steps > 3 * triples:
                    run(bj, sig, ta, th, 3 * triples,
                        args.steps - 3 * triples, triples == 0)
                ops.scale(x, scales[j], out=results[j * nvox:(j + 1) * nvox])
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            if world > 1:
                bucket = gather_results(results)
                del bucket
            gathers.append((time.perf_counter() - g0) * 1e3)
            fence()
            rep_s.append(max_over_ranks(time.perf_counter() - t0))
            if triples:
                kms = ev.elapsed_ms(e0, em) / triples
            rep_kernel_ms.append(kms)
        gather_ms = median(gathers) if world > 1 else None

    tmed = median(rep_s)
    kernel_ms = median(rep_kernel_ms)
    # every rank's own figure travels to rank 0: a straggling GPU shows in the line
    per_rank_ms = [kernel_ms]
    if world > 1:
        mine_ms = torch.tensor([kernel_ms], dtype=torch.float64,
                               device="cpu" if on_host else dev)
        every = [torch.empty_like(mine_ms) for _ in range(world)]
        dist.all_gather(every, mine_ms)
        per_rank_ms = [float(t.item()) for t in every]
    if triples:
        launches, iters_per_launch = triples, 3
    else:
        launches, iters_per_launch = 1, args.steps

    # The timed state proves itself: replay the SAME schedule from the same
    # initial state through the one-iteration-per-launch kernel (k_pd_fused,
    # the form held to the reference goldens and the oracle) and require the
    # final x, xbar and p to be bit-identical to what the timed launches left
    # (primal_dual_solver.py:242-256 per iteration).  Every rank checks its own
    # volume; in --batch mode the last volume of the rank's share.
    verify = None
    if not args.no_verify:
        a = state["slot"]
        timed = (x.clone(), xbar[a].clone(), p[a].clone())
        launches_k = ops.pd_fusedk_launches(3)
        _lib.set_param("pd2_enable", 0)
        _lib.set_param("pdk_enable", 0)
        try:
            if not args.batch:
                reset(bt)
                if args.warmup > 0:
                    run(bt, sig, ta, th, 0, args.warmup, True)
                first = args.warmup
                for r in range(reps):
                    if triples:
                        run(bt, sig, ta, th, first, 3 * triples, first == 0)
                    if args.steps > 3 * triples:
                        run(bt, sig, ta, th, first + 3 * triples,
                            args.steps - 3 * triples,
                            first == 0 and triples == 0)
                    first += args.steps
                replayed = total
            else:
                reset(inputs[-1])
                run(inputs[-1], sig, ta, th, 0, args.steps, True)
                replayed = args.steps
            torch.cuda.synchronize()
        finally:
            _lib.set_param("pd2_enable", 1)
            _lib.set_param("pdk_enable", 1)
        a = state["slot"]
        same = [bool(torch.equal(u, v)) for u, v in
                zip(timed, (x, xbar[a], p[a]))]
        ok = torch.tensor([int(all(same))], dtype=torch.int64,
                          device="cpu" if on_host else dev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        verify = {"bit_identical": bool(ok.item()),
                  "rank0_arrays": dict(zip(("x", "xbar", "p"), same)),
                  "iterations_replayed": replayed,
                  "depth3_launches_during_replay":
                      ops.pd_fusedk_launches(3) - launches_k,
                  "how": "same schedule from the same initial state, one "
                         "launch of k_pd_fused per iteration (pdk_enable=0, "
                         "pd2_enable=0); torch.equal on x, xbar and p; every "
                         "rank, MIN over ranks"}
        del timed

    # for reference: the one-iteration-per-pass kernel on the same state
    single = None
    if world == 1 and args.steps >= 2 and not args.batch:
        _lib.set_param("pd2_enable", 0)
        _lib.set_param("pdk_enable", 0)
        try:
            k1 = 40
            sg1, ta1, th1 = step_schedule("ALG2", 16.0, lmbda, k1)
            f0, f1 = ev.create(), ev.create()
            ops.pd_run(xbar[0], xbar[1], x_alt, bt, p[0], p[1], shape, w,
                       lmbda, sg1[:4], ta1[:4], th1[:4], False, 0.05, flags)
            ev.record(f0, stream)
            ops.pd_run(xbar[0], xbar[1], x_alt, bt, p[0], p[1], shape, w,
                       lmbda, sg1, ta1, th1, False, 0.05, flags)
            ev.record(f1, stream)
            torch.cuda.synchronize()
            ms1 = ev.elapsed_ms(f0, f1) / k1
            single = {"kernel": "k_pd_fused (one iteration per launch)",
                      "avg_launch_ms": ms1, "iterations_per_s": 1e3 / ms1,
                      "achieved": BYTES_PER_VOXEL * nvox / (ms1 * 1e-3) / 1e9,
                      "frac": BYTES_PER_VOXEL * nvox / (ms1 * 1e-3) / 1e9 /
                      HBM_PEAK_GBPS}
        finally:
            _lib.set_param("pd2_enable", 1)
            _lib.set_param("pdk_enable", 1)

    finite = bool(torch.isfinite(x).all().item())
    joined = torch.ones(1, dtype=torch.int64, device="cpu" if on_host else dev)
    if world > 1:
        dist.all_reduce(joined)
    if int(joined.item()) != world:
        sys.stderr.write("bench.py: %d of %d ranks joined\n"
                         % (int(joined.item()), world))
        return 3

    if rank == 0:
        volumes = args.batch or world
        value = volumes * args.steps / tmed
        # algorithmic bytes: 44 B per voxel per ITERATION; a launch
        # of the dominant kernel processes iters_per_launch iterations ...
        bytes_per_launch = BYTES_PER_VOXEL * nvox * iters_per_launch
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        # ... while reading and writing the eleven arrays ONCE: the least a
        # launch can physically move
        pass_bytes = BYTES_PER_VOXEL * nvox
        traffic, traffic_source = profiled_traffic(n, plan)
        out = {
            "metric": "primal-dual iters/sec on %d^3 fp32 TV-%s" % (n, data),
            "value": value, "unit": "iterations/s (summed over volumes)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": tmed / args.steps * 1e3 *
            (world / float(volumes)),
            "higher_is_better": True,
            "scaling": "strong" if args.batch else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": (
                    "batch of %d x synth_volume(%d, seed=i, '%s'), TV-%s "
                    "denoising, Chambolle-Pock ALG2, L2=16, alpha=%g, %d "
                    "iterations per volume, volume i -> rank i %% %d, one "
                    "gather at the end inside the clock (BASELINE config 5)"
                    % (args.batch, n, kind, data, alpha, args.steps, world))
                if args.batch else (
                    "synth_volume(%d, seed=rank, '%s'), TV-%s "
                    "denoising, Chambolle-Pock ALG2, L2=16, "
                    "alpha=%g, one volume per GPU" % (n, kind, data, alpha)),
                "volumes": volumes, "voxels_per_volume": nvox,
                "kernel": "k_pd_fusedk (three iterations per pass; trailing "
                          "iterations: k_pd_fused2 / k_pd_fused)",
                "kernel_config": None if plan is None else {
                    "waves": plan[0], "tiles_x": plan[1], "zchunk": plan[2],
                    "pinned": bool(pinned)},
                "timed_repetitions": reps,
                "repetition_ms": [t * 1e3 for t in rep_s],
                "statistic": "median of the repetitions (each: barrier + "
                             "synchronize, K steps, synchronize + barrier; "
                             "max over ranks)",
                "backend": None if world == 1 else
                ("rccl" if args.backend == "nccl" else "gloo (host staging)"),
                "ranks_joined": int(joined.item()),
                "gather_ms": gather_ms,
                "gather_inside_clock": bool(args.batch),
                "result_finite": finite,
                "result_bit_identical_to_single_pass":
                    None if verify is None else verify["bit_identical"],
                "verification": verify},
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_source": traffic_source,
                # `frac` prices iters_per_launch iterations at 44 B per voxel
                # each (the contract's algorithmic bytes) although the launch
                # moves the arrays once -- it may exceed 1.  The two fractions
                # below are against what a launch physically has to move / did
                # move:
                "frac_per_launch": pass_bytes / (kernel_ms * 1e-3) / 1e9 /
                HBM_PEAK_GBPS,
                "frac_traffic": (traffic / (kernel_ms * 1e-3) / 1e9 /
                                 HBM_PEAK_GBPS) if traffic else None,
                "bytes_per_launch": bytes_per_launch,
                "min_physical_bytes_per_launch": pass_bytes,
                "iterations_per_launch": iters_per_launch,
                "launches": launches,
                "avg_launch_ms": kernel_ms,
                "avg_launch_ms_per_rank": per_rank_ms,
                "avg_launch_ms_repetitions": rep_kernel_ms,
                "single_pass_reference": single},
        }
        if world == 1 and not args.no_cpu_baseline and not args.batch:
            sn = args.cpu_sample or n
            its = cpu_baseline(sn, 2)
            out["cpu_baseline"] = {
                "value": its * (sn ** 3) / float(nvox),
                "unit": "iterations/s", "cores": 1, "kind": "port",
                "host_cores_available": os.cpu_count(),
                "sample": "oracle pd_tvl2_refstyle (NumPy float64 + "
                          "scipy.ndimage, reference op sequence), %d^3 "
                          "volume, 2 iterations after 1 warm-up%s" %
                          (sn, "" if sn == n else
                           ", scaled by voxel count to %d^3" % n)}
            try:
                itc, threads = cpu_baseline_c(sn, 6)
                out["cpu_baseline_compiled"] = {
                    "value": itc * (sn ** 3) / float(nvox),
                    "unit": "iterations/s", "cores": threads, "kind": "port",
                    "sample": "oracle/pd_oracle.c (C + OpenMP float64, "
                              "bit-identical to the NumPy oracle), %d^3 "
                              "volume, 6 iterations after 1 warm-up" % sn}
            except Exception as e:             # no gcc on the box: say so
                out["cpu_baseline_compiled"] = {"error": str(e)}
        if world == 1 and not args.batch and not args.no_config4 and n >= 64:
            # (the solver state of the headline run is no longer needed)
            del p[:], xbar[:], inputs[:]
            torch.cuda.empty_cache()
            try:
                out["config4"] = config4_line(
                    n, cpu_sample=0 if args.no_cpu_baseline else 64)
            except Exception as e:                  # never at the headline's expense
                out["config4"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if verify is not None and not verify["bit_identical"]:
        sys.stderr.write("bench.py: the timed state is NOT bit-identical to "
                         "the one-iteration-per-launch replay: %r\n" % (verify,))
        return 4
    return 0


if __name__ == "__main__":
    sys.exit(main())
