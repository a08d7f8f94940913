#!/bin/bash
# Copies what tools/battery.sh <tag> left under gpurun_out/ into profiles/<tag>_* (the
# tracked summaries), replacing the set of <old tag> and the references to it in the docs.
#   tools/install_profiles.sh <tag> [<old tag>]
set -e
T=$1; OLD=$2
CFG=$(cat gpurun_out/${T}_config)
python tools/summarize_profiles.py --tag $T --stats gpurun_out/${T}_stats --fetch gpurun_out/${T}_fetch --write gpurun_out/${T}_write --config $CFG --last 10 > /dev/null
for f in bench.json bench_driver_args.json bench_admm_lsmr.json bench_admm_lbfgsb_huber.json bench_admm_lsmr_511.json bench_admm_lbfgsb_huber_511.json bench_pd_deconv.json bench_2ranks_gloo.json bench_batch4_2ranks_gloo.json bench_batch8.json bench_f64.jsonl bench_persist.jsonl bench_shapes.jsonl bench_small.jsonl blur3_ragged.jsonl blur3_taps.jsonl lbfgsb_kernels.jsonl parity_errors.json pytest_gpu.log smoke.log; do cp gpurun_out/${T}_$f profiles/${T}_$f; done
python3 tools/summarize_pmc.py k_ gpurun_out/${T}_admm_pmc_fetch gpurun_out/${T}_admm_pmc_write > profiles/${T}_admm_pmc.jsonl
cp profiles/${T}_admm_pmc.jsonl profiles/latest_admm_pmc.jsonl   # what bench_admm.py cites
python3 tools/summarize_pmc.py k_ gpurun_out/${T}_huber_pmc_fetch gpurun_out/${T}_huber_pmc_write > profiles/${T}_admm_lbfgsb_huber_pmc.jsonl
python3 tools/summarize_pmc.py blur3 gpurun_out/${T}_blur3_stats gpurun_out/${T}_blur3_fetch gpurun_out/${T}_blur3_write gpurun_out/${T}_blur3_sq1 gpurun_out/${T}_blur3_sq2 > profiles/${T}_blur3_pmc.jsonl
for d in admm_lsmr admm_lbfgsb pd_deconv blur3_stats; do f=$(find gpurun_out/${T}_$d -name "p_kernel_stats.csv" | head -1); cp $f profiles/${T}_${d}_kernel_stats.csv; done
mv profiles/${T}_admm_lbfgsb_kernel_stats.csv profiles/${T}_admm_lbfgsb_huber_kernel_stats.csv
mv profiles/${T}_blur3_stats_kernel_stats.csv profiles/${T}_blur3_kernel_stats.csv
grep '"metric"' gpurun_out/${T}_stats.log | tail -1 > profiles/${T}_bench_profiled.json
if [ -f gpurun_out/${T}_huber_run_trace.txt ]; then
  cp gpurun_out/${T}_huber_run_trace.txt profiles/${T}_huber_run_trace.txt
  cp gpurun_out/${T}_huber_trace/p_kernel_stats.csv profiles/${T}_huber_run_kernel_stats.csv
fi
if [ -n "$OLD" ]; then
  git rm -q --cached profiles/${OLD}_* 2>/dev/null || true
  rm -f profiles/${OLD}_*
  sed -i "s/${OLD}_/${T}_/g" DESIGN.md README.md
fi
ls profiles | grep -c $T
