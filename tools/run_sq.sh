#!/bin/bash
# The two SQ-counter passes of tools/run_counters.sh alone (no traffic passes, no trace).
# Usage: bash tools/run_sq.sh <tag> <workload_tag> <bench.py flags...>
set -e
tag=$1
wl=$2
shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
bench=$GRAFT_REPO_ROOT/bench.py
cd /tmp && export TMPDIR=/tmp
SQ_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE"
SQ_B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
timeout -s KILL 240 rocprofv3 --pmc $SQ_A --output-format csv -d $out/sq_a -- python3 $bench --no-cpu-baseline --no-kernel-timer --no-pointset --no-side-workloads --steps 5 "$@" > $out/sq_a.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc $SQ_B --output-format csv -d $out/sq_b -- python3 $bench --no-cpu-baseline --no-kernel-timer --no-pointset --no-side-workloads --steps 5 "$@" > $out/sq_b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/sq_summary.py $out/sq_a $out/sq_b --workload $wl > $out/sq.json
rm -rf $out/sq_a $out/sq_b
python - <<PY
import json
k = json.load(open("$out/sq.json"))["kernels"]
for name, v in k.items():
    r = v["ratios"]
    print("%-50s n=%3d %s" % (name[:50], v["launches_sampled"], " ".join("%s=%.3f" % kv for kv in r.items())))
PY
