#!/bin/bash
# GPU-side durations of the fully connected launches (tools/fc_bench.py under a rocprofv3
# kernel trace: HIP-event timing of single small launches measures the host's launch rate).
# Usage: bash tools/fc_trace.sh <tag> [rows]
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/fc_bench.py ${2:-1024} > $out/fc_bench.txt 2>&1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
grep -i "fc_stack\|fewrow\|shared_mlp" $out/kernel_stats.csv | cut -c1-200
