#!/bin/bash
# Hardware-counter record of one bench.py workload: HBM traffic (FETCH_SIZE, WRITE_SIZE, one
# pass each) and two passes of SQ counters (matrix-core / vector-ALU / LDS activity and the
# wave's wait states) plus a rocprofv3 kernel trace of the same command.
# Usage: bash tools/run_counters.sh <tag> <workload_tag> <bench.py flags...>
# Output: gpurun_out/<tag>/{pmc_traffic.json, sq.json, kernel_stats.csv, bench.json}
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
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $bench --no-cpu-baseline --no-kernel-timer "$@" > $out/pmc_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $bench --no-cpu-baseline --no-kernel-timer "$@" > $out/pmc_write.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc $SQ_A --output-format csv -d $out/sq_a -- python3 $bench --no-cpu-baseline --no-kernel-timer "$@" > $out/sq_a.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc $SQ_B --output-format csv -d $out/sq_b -- python3 $bench --no-cpu-baseline --no-kernel-timer "$@" > $out/sq_b.log 2>&1
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $bench --no-cpu-baseline "$@" > $out/bench.json 2> $out/trace.log
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write --workload $wl > $out/pmc_traffic.json
python tools/sq_summary.py $out/sq_a $out/sq_b --workload $wl > $out/sq.json
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/pmc_fetch $out/pmc_write $out/sq_a $out/sq_b $out/trace      # raw per-dispatch records: tens of MiB
python - <<PY
import json
k = json.load(open("$out/sq.json"))["kernels"]
for name, v in k.items():
    r = v["ratios"]
    print("%-60s n=%3d %s" % (name[:60], v["launches_sampled"], " ".join("%s=%.3f" % kv for kv in r.items())))
PY
