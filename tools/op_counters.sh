#!/bin/bash
# Two passes of SQ counters over any program of this repository (a tools/*.py that launches a few kernels).
# Usage: bash tools/op_counters.sh <tag> <script.py> [args...]      -> gpurun_out/<tag>/sq.json
set -e
tag=$1
shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
prog=$GRAFT_REPO_ROOT/$1
shift
cd /tmp && export TMPDIR=/tmp
SQ_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE"
SQ_B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
SQ_C="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --pmc $SQ_A --output-format csv -d $out/sq_a -- python3 $prog "$@" > $out/sq_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc $SQ_B --output-format csv -d $out/sq_b -- python3 $prog "$@" > $out/sq_b.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc $SQ_C --output-format csv -d $out/sq_c -- python3 $prog "$@" > $out/sq_c.log 2>&1 || echo "third pass failed (see sq_c.log)"
cd $GRAFT_REPO_ROOT
python tools/sq_summary.py $out/sq_a $out/sq_b $out/sq_c --by-grid > $out/sq.json
rm -rf $out/sq_a $out/sq_b $out/sq_c
python - <<PY
import json
k = json.load(open("$out/sq.json"))["kernels"]
for name, v in k.items():
    r = v["ratios"]
    print("%-70s n=%3d %s" % (name[:70], v["launches_sampled"], " ".join("%s=%.3f" % kv for kv in r.items())))
    print("      ", {a: int(b) for a, b in v["counters"].items()})
PY
