#!/bin/bash
# bf16 measurements: v2 at both precisions, v1 bf16, and the rocprofv3 kernel statistics
# of the v1 bf16 run.  Output under gpurun_out/r01_f/.
set -e
out=gpurun_out/r01_f
mkdir -p $out
timeout -k 10 200 python -u bench.py --config v2 > $out/bench_v2_f32.json 2> $out/bench_v2_f32.err
timeout -k 10 200 python -u bench.py --config v2 --dtype bf16 > $out/bench_v2_bf16.json 2> $out/bench_v2_bf16.err
timeout -k 10 200 python -u bench.py --dtype bf16 --no-cpu-baseline > $out/bench_v1_bf16.json 2> $out/bench_v1_bf16.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o v1_bf16 -- python3 $GRAFT_REPO_ROOT/bench.py --dtype bf16 --no-cpu-baseline --no-pointset --no-kernel-timer > $GRAFT_REPO_ROOT/$out/prof_run.log 2>&1
cd $GRAFT_REPO_ROOT
tail -c 600 $out/bench_v2_f32.json; echo; tail -c 600 $out/bench_v2_bf16.json; echo; head -c 900 $out/bench_v1_bf16.json; echo
find $out/prof -name "*kernel_stats*" | head
