#!/bin/bash
# Counter + trace records of every bench.py workload (see tools/run_counters.sh), then the
# driver's own command lines with the CPU baselines.  Usage: bash tools/run_r02_counters_all.sh <prefix>
set -e
pre=${1:-r02e}
cd $GRAFT_REPO_ROOT
bash tools/run_counters.sh ${pre}_v1 v1_f32_1024x1024 --steps 40 --no-side-workloads > gpurun_out/${pre}_v1.txt
bash tools/run_counters.sh ${pre}_v2_f32 v2_f32_1024x2048 --config v2 --steps 5 --no-pointset > gpurun_out/${pre}_v2_f32.txt
bash tools/run_counters.sh ${pre}_v2_bf16 v2_bf16_1024x2048 --config v2 --dtype bf16 --steps 5 --no-pointset > gpurun_out/${pre}_v2_bf16.txt
bash tools/run_counters.sh ${pre}_bayes_f32 bayes_f32_S32_128x1024 --config bayes --steps 2 --warmup 1 > gpurun_out/${pre}_bayes_f32.txt
bash tools/run_counters.sh ${pre}_bayes_bf16 bayes_bf16_S32_128x1024 --config bayes --dtype bf16 --steps 2 --warmup 1 > gpurun_out/${pre}_bayes_bf16.txt
echo counters done
timeout -k 10 300 python3 bench.py --config v2 > gpurun_out/${pre}_bench_v2_full.json 2> gpurun_out/${pre}_bench_v2_full.err
timeout -k 10 300 python3 bench.py --config bayes --steps 3 > gpurun_out/${pre}_bench_bayes_full.json 2> gpurun_out/${pre}_bench_bayes_full.err
echo benches done
