#!/bin/bash
# rocprofv3 kernel statistics of the Flipout workload.  Output under gpurun_out/r01_g/.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/r01_g
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bayes -- python3 $GRAFT_REPO_ROOT/bench.py --config bayes --steps 3 --warmup 1 > $out/prof_run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-200
