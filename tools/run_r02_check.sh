#!/bin/bash
# Round check: counter list, bench lines of the three workloads, a 2-rank gloo rehearsal.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $out
cd $GRAFT_REPO_ROOT
(timeout -k 10 60 rocprofv3 -L > $out/counters.txt 2>&1 || true)
timeout -k 10 300 python3 bench.py > $out/bench_v1.json 2> $out/bench_v1.err
timeout -k 10 300 python3 bench.py --config v2 > $out/bench_v2.json 2> $out/bench_v2.err
timeout -k 10 300 python3 bench.py --config v2 --dtype bf16 > $out/bench_v2_bf16.json 2> $out/bench_v2_bf16.err
timeout -k 10 300 python3 bench.py --config bayes --steps 3 > $out/bench_bayes.json 2> $out/bench_bayes.err
timeout -k 10 300 python3 bench.py --config bayes --dtype bf16 --steps 3 > $out/bench_bayes_bf16.json 2> $out/bench_bayes_bf16.err
FPN_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 2 --warmup 1 --batch 64 > $out/bench_gloo2.json 2> $out/bench_gloo2.err
FPN_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 1 --warmup 1 --config bayes --batch 8 --samples 4 > $out/bench_gloo2_bayes.json 2> $out/bench_gloo2_bayes.err
echo done
