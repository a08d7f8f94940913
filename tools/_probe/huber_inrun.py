#!/usr/bin/env python3
"""Config 4 (L-BFGS-B / Huber branch) seconds per run and the in-run totals of its main
entries (bench_admm.measure); one line, for A/B runs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench_admm
r = bench_admm.measure(int(sys.argv[1]) if len(sys.argv) > 1 else 512, minimizer="L-BFGS-B",
                       data_loss="huber", repeat=4, cpu_sample=0)
k = r["roofline"]["kernels"]
tot = lambda pre: sum(v["ms_per_run"] for n, v in k.items() if n.startswith(pre))
print("s/run %.4f  kernels %.1f ms  idle %.1f ms | objective %.1f  evalblurs %.1f  cauchy %.1f (finish %.2f ms each)  gram %.1f  step %.1f  mdots %.1f  vw %.1f" % (
    r["seconds_per_run"], r["roofline"]["timed_run_kernel_ms"], r["roofline"]["timed_run_gpu_idle_ms"],
    tot("tk1_reg_objective"), tot("corr3_wrap"), tot("lb_cauchy") + tot("lb_select") + tot("lb_sort"),
    k["lb_cauchy_finish"]["avg_launch_ms"], tot("lb_masked_gram"), tot("lb_subspace_step"), tot("lb_mdots"),
    tot("admm_vw")), flush=True)
