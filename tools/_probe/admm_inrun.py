#!/usr/bin/env python3
"""One line: config 4 (LSMR branch) seconds per run and the in-run durations of the two
Lanczos halves (bench_admm.measure); for A/B runs of library builds."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench_admm
r = bench_admm.measure(int(sys.argv[1]) if len(sys.argv) > 1 else 512, repeat=4, cpu_sample=0)
k = r["roofline"]["kernels"]
print("s/run %.4f  a2 %.4f ms  b2 %.4f ms  vw %.4f  wcomb %.4f" % (
    r["seconds_per_run"], k["corr3_wrap_lanczos_a2"]["avg_launch_ms"],
    k["corr3_wrap_lanczos_b2"]["avg_launch_ms"], k["admm_vw_update_norm"]["avg_launch_ms"],
    k["lincomb_clip#10"]["avg_launch_ms"]), flush=True)
