#!/bin/bash
# The round's measurements of the headline workload (python bench.py, fp32): the bench
# line, rocprofv3 kernel statistics of the same command, and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE; each in a run of its own) that tools/pmc_summary.py turns into
# HBM bytes per launch.  Usage: bash tools/run_round_profile.sh <tag>   (output under
# gpurun_out/<tag>/)
set -e
tag=${1:-round}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
args="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pointset --no-kernel-timer --no-side-workloads --steps 5 --warmup 2"
# (40 steps: the first launches after start-up run at ramping clocks and would dominate a short average;
# the headline workload alone: the side workloads launch the same kernels with per-copy weights)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o v1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pointset --no-kernel-timer --no-side-workloads --steps 40 --warmup 5 > $out/stats_run.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $args > $out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $args > $out/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write --workload v1_f32_1024x1024 > $out/pmc_traffic.json
head -c 700 $out/bench.json; echo
head -8 $(find $out/stats -name "*kernel_stats.csv" | head -1) | cut -c1-170
head -c 1500 $out/pmc_traffic.json
