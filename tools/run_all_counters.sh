# Hardware-counter records of all six bench.py workloads (tools/run_counters.sh each).
# Usage: bash tools/run_all_counters.sh <tag prefix> [first|second]   (half of the workloads: a call's time limit)
set -e
cd $GRAFT_REPO_ROOT
t=$1
half=${2:-both}
if [ "$half" != second ]; then
bash tools/run_counters.sh ${t}_v1 v1_f32_1024x1024 --no-pointset --no-side-workloads --steps 10 --warmup 3 > gpurun_out/${t}_v1.log 2>&1
echo v1 done
bash tools/run_counters.sh ${t}_v1_bf16 v1_bf16_1024x1024 --no-pointset --no-side-workloads --dtype bf16 --steps 10 --warmup 3 > gpurun_out/${t}_v1_bf16.log 2>&1
echo v1bf16 done
bash tools/run_counters.sh ${t}_bayes_bf16 bayes_bf16_S32_128x1024 --no-pointset --config bayes --dtype bf16 --steps 3 --warmup 1 > gpurun_out/${t}_bayes_bf16.log 2>&1
echo bayesbf16 done
fi
if [ "$half" != first ]; then
bash tools/run_counters.sh ${t}_v2_f32 v2_f32_1024x2048 --no-pointset --config v2 --steps 5 --warmup 2 > gpurun_out/${t}_v2_f32.log 2>&1
echo v2 done
bash tools/run_counters.sh ${t}_v2_bf16 v2_bf16_1024x2048 --no-pointset --config v2 --dtype bf16 --steps 5 --warmup 2 > gpurun_out/${t}_v2_bf16.log 2>&1
echo v2bf16 done
bash tools/run_counters.sh ${t}_bayes_f32 bayes_f32_S32_128x1024 --no-pointset --config bayes --steps 3 --warmup 1 > gpurun_out/${t}_bayes_f32.log 2>&1
echo bayes done
fi
