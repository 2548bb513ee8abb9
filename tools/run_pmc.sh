#!/bin/bash
# HBM traffic per kernel (rocprofv3 PMC: FETCH_SIZE and WRITE_SIZE, one pass each) of any
# bench.py workload.  Usage: bash tools/run_pmc.sh <tag> <bench.py flags...>
# Output: gpurun_out/<tag>/pmc_traffic.json (keys "kernel @ grid size")
set -e
tag=$1
shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $out/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write --by-grid > $out/pmc_traffic.json
python - <<PY
import json
k = json.load(open("$out/pmc_traffic.json"))["kernels"]
for name, v in sorted(k.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:24]:
    print("%-66s n=%3d fetch x2 %9.1f MB  write %9.1f MB" % (name[:66], v["launches_sampled"], v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6))
PY
