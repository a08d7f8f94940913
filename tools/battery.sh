#!/bin/bash
# The measurement battery behind profiles/<tag>_*: GPU test suite (observed
# errors), smoke, headline bench (plain, under rocprofv3 --stats, PMC passes),
# config 4 (both branches, plain and under rocprofv3 --stats), the blur's PMC
# passes, the driver's own bench invocation, the launcher rehearsals.
#   tools/battery.sh <tag> [a|b|all]   (on the GPU box; results under gpurun_out/;
#   two halves, each within one 20-minute gpurun call)
set -e
TAG=${1:-prof}
PART=${2:-all}
O=gpurun_out
mkdir -p $O
if [ "$PART" != "b" ]; then
timeout -k 10 500 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/${TAG}_pytest_gpu.log
cp $O/parity_errors.json $O/${TAG}_parity_errors.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1
timeout -k 10 500 bash tools/profile_bench.sh ${TAG}
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_args.json 2>> $O/${TAG}_bench.err
timeout -k 10 300 bash tools/profile_admm.sh ${TAG}
python bench_admm.py > $O/${TAG}_bench_admm_lsmr.json 2>/dev/null
python bench_admm.py --minimizer L-BFGS-B --data-loss huber > $O/${TAG}_bench_admm_lbfgsb_huber.json 2>/dev/null
fi
if [ "$PART" = "a" ]; then echo BATTERY_A_DONE; exit 0; fi
timeout -k 10 300 bash tools/profile_blur3.sh ${TAG}
python3 tools/summarize_pmc.py blur3 $O/${TAG}_blur3_stats $O/${TAG}_blur3_fetch $O/${TAG}_blur3_write $O/${TAG}_blur3_sq1 $O/${TAG}_blur3_sq2 > $O/${TAG}_blur3_pmc.jsonl
python bench.py --gpus 2 --backend gloo --steps 300 > $O/${TAG}_bench_2ranks_gloo.json 2>> $O/${TAG}_bench.err
python bench.py --gpus 2 --backend gloo --batch 4 --steps 150 > $O/${TAG}_bench_batch4_2ranks_gloo.json 2>> $O/${TAG}_bench.err
python bench.py --batch 8 --steps 150 --no-cpu-baseline > $O/${TAG}_bench_batch8.json 2>> $O/${TAG}_bench.err
python tools/bench_small.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_small.jsonl
python tools/bench_persist.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_persist.jsonl
python tools/bench_lbfgsb_kernels.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_lbfgsb_kernels.jsonl
python tools/bench_blur3_taps.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_blur3_taps.jsonl
python tools/bench_shapes.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_shapes.jsonl
# rows / lengths that are not whole vectors: the blur, config 4 at 511^3
(python tools/bench_blur3.py 511 2>/dev/null | grep -v amdgpu | head -2; python tools/bench_blur3.py 512,512,509 2>/dev/null | grep -v amdgpu | head -2) > $O/${TAG}_blur3_ragged.jsonl
python bench_admm.py --size 511 --no-cpu-baseline --repeat 3 > $O/${TAG}_bench_admm_lsmr_511.json 2>/dev/null
python bench_admm.py --size 511 --no-cpu-baseline --repeat 3 --minimizer L-BFGS-B --data-loss huber > $O/${TAG}_bench_admm_lbfgsb_huber_511.json 2>/dev/null
# primal-dual deconvolution (plain and under rocprofv3 --stats)
timeout -k 10 300 bash tools/profile_pd_deconv.sh ${TAG}
# the reference's own arithmetic type
python tools/bench_f64.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_f64.jsonl
# PMC traffic of config 4's kernels (LSMR branch)
timeout -k 10 400 bash tools/profile_admm_pmc.sh ${TAG}
python3 tools/summarize_pmc.py k_ $O/${TAG}_admm_pmc_fetch $O/${TAG}_admm_pmc_write > $O/${TAG}_admm_pmc.jsonl
# ... and of the L-BFGS-B / Huber branch
timeout -k 10 400 bash tools/profile_admm_pmc_huber.sh ${TAG}
python3 tools/summarize_pmc.py k_ $O/${TAG}_huber_pmc_fetch $O/${TAG}_huber_pmc_write > $O/${TAG}_admm_lbfgsb_huber_pmc.jsonl
# one Huber run's kernel list and the GPU's idle time in it
bash tools/_probe/huber_run_trace.sh ${TAG} > /dev/null 2>&1
(grep -E "^run|sum nit|cauchy" $O/${TAG}_huber_trace.log; python3 tools/_probe/trace_gaps.py $O/${TAG}_huber_trace/p_kernel_trace.csv 3 -v | tail -34) > $O/${TAG}_huber_run_trace.txt
echo BATTERY_DONE
