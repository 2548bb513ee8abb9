# counter records of the two bf16 v1-topology workloads (tag prefix $1)
set -e
cd $GRAFT_REPO_ROOT
bash tools/run_counters.sh ${1}_v1_bf16 v1_bf16_1024x1024 --no-pointset --dtype bf16 --steps 10 --warmup 3 > gpurun_out/${1}_v1_bf16.log 2>&1
bash tools/run_counters.sh ${1}_bayes_bf16 bayes_bf16_S32_128x1024 --no-pointset --config bayes --dtype bf16 --steps 3 --warmup 1 > gpurun_out/${1}_bayes_bf16.log 2>&1
timeout -k 10 300 python -u bench.py > gpurun_out/${1}_bench_default.json 2> gpurun_out/${1}_bench_default.err
tail -c 600 gpurun_out/${1}_bench_default.json
