#!/bin/bash
# rocprofv3 kernel statistics of the v2 workload at both precisions (gpurun_out/r01_i/).
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/r01_i
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for dt in f32 bf16; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$dt -o v2 -- python3 $GRAFT_REPO_ROOT/bench.py --config v2 --dtype $dt --steps 5 --warmup 2 > $out/prof_$dt.log 2>&1
done
cd $GRAFT_REPO_ROOT
for dt in f32 bf16; do echo == $dt; tail -1 $out/prof_$dt.log | cut -c1-200; head -14 $(find $out/prof_$dt -name "*kernel_stats.csv" | head -1) | cut -d, -f1-5 | cut -c1-140; done
