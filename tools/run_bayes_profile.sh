#!/bin/bash
# rocprofv3 kernel statistics of the Flipout workload.  Usage: run_bayes_profile.sh <tag> [bench flags]
# Output under gpurun_out/<tag>/.
set -e
tag=${1:-bayes}
shift || true
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bayes -- python3 $GRAFT_REPO_ROOT/bench.py --config bayes --steps 3 --warmup 1 "$@" > $out/prof_run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $out/kernel_stats.csv
head -25 "$f" | cut -c1-200
