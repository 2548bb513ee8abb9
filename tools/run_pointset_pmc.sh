#!/bin/bash
# HBM traffic (rocprofv3 PMC: FETCH_SIZE and WRITE_SIZE, one pass each) of the point-set
# operators as bench.py's `pointset_ops` section launches them (256 frustums, 2048 points,
# 128 samples, 64 per ball, 64 channels).  Usage: bash tools/run_pointset_pmc.sh <tag>
set -e
tag=${1:-pointset}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export FPN_POINTSET_SYNC=1     # one launch on the device at a time
args="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timer --steps 1 --warmup 1"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $args > $out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $args > $out/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write --by-grid > $out/pointset_pmc_traffic.json
python - <<PY
import json
k = json.load(open("$out/pointset_pmc_traffic.json"))["kernels"]
for name, v in k.items():
    if any(s in name for s in ("fps", "query_ball", "group", "three_nn", "gather_point")):
        print("%-60s fetch raw %12.0f  x2 %12.0f  write %12.0f" % (name[:60], v["fetch_bytes_raw"], v["fetch_bytes"], v["write_bytes"]))
PY
