#!/usr/bin/env python3
"""Config 4 (LSMR branch) seconds per run and the in-run durations of its main entries
(bench_admm.measure), one line per setting; for A/B runs of library builds and host
switches:  admm_inrun.py [size] [module.NAME=value ...]   (each switch: a second line)"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench_admm
args = sys.argv[1:]
n = int(args.pop(0)) if args and args[0].isdigit() else 512


def line(tag):
    r = bench_admm.measure(n, repeat=4, cpu_sample=0)
    k = r["roofline"]["kernels"]
    get = lambda name: k[name]["avg_launch_ms"] if name in k else float("nan")
    print("%-28s s/run %.4f  a2 %.4f  b2 %.4f  vw %.4f  vw_g %.4f  v %.4f  wcomb %.4f  idle %.1f ms" % (
        tag, r["seconds_per_run"], get("corr3_wrap_lanczos_a2"), get("corr3_wrap_lanczos_b2"),
        get("admm_vw_update_norm"), get("admm_vw_update_g"), get("lsmr_v_update_to"),
        get("lincomb_clip#10"), r["roofline"]["timed_run_gpu_idle_ms"]), flush=True)


line("default")
for kv in args:
    path, v = kv.split("=")
    mod, name = path.rsplit(".", 1)
    m = importlib.import_module("nsol_amd." + mod)
    old = getattr(m, name)
    setattr(m, name, type(old)(int(v)))
    line(kv)
    setattr(m, name, old)
