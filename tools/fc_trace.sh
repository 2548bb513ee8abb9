#!/bin/bash
# GPU-side durations of the fully connected launches (tools/fc_bench.py under a rocprofv3
# kernel trace: HIP-event timing of single small launches measures the host's launch rate),
# per kernel and grid size.  Usage: bash tools/fc_trace.sh <tag> [rows]
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/fc_bench.py ${2:-1024} > $out/fc_bench.txt 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for path in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "fpn::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("fpn::(anonymous namespace)::", "").split("(")[0]
            agg[(name, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, grid), v in sorted(agg.items()):
    v.sort()
    print("%-40s grid %8s  n=%4d  median %7.1f us  min %7.1f" % (name[:40], grid, len(v), v[len(v) // 2] / 1e3, v[0] / 1e3))
PY
rm -rf $out/trace
