#!/bin/bash
# Timeline of one step of a bench workload from a rocprofv3 kernel trace: every launch of
# the LAST timed step in order (duration, idle gap before it), then the per-kernel totals
# of that step.  Usage: bash tools/step_trace.sh <tag> <bench flags...>
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py "$@" \
    --steps 6 --warmup 1 --no-kernel-timer --no-cpu-baseline --no-pointset --no-side-workloads > $out/bench.json 2> $out/bench.err
python3 - $out <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
rows = []
for path in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(n):
    n = n.replace("fpn::(anonymous namespace)::", "").replace("void ", "")
    n = n.replace("at::native::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]


# steps are delimited by the first kernel of the model: the launches between the last two
# occurrences of the most frequent fpn kernel sequence start (mask_gather runs once a step)
marks = [i for i, r in enumerate(rows) if "mask_gather_kernel" in r[2]]
if len(marks) < 3:
    sys.exit("too few steps in the trace")
per = marks[-1] - marks[-2]
lo, hi = marks[-2], marks[-1]
step = rows[lo:hi]
span = step[-1][1] - step[0][0]
busy = sum(e - s for s, e, _ in step)
with open(out + "/timeline.txt", "w") as f:
    f.write("one step (mask_gather to mask_gather): %d launches, span %.1f us, busy %.1f us\n" % (per, span / 1e3, busy / 1e3))
    prev = None
    for s, e, n in step:
        f.write("%8.1f us  gap %6.1f  %s\n" % ((e - s) / 1e3, 0 if prev is None else (s - prev) / 1e3, short(n)))
        prev = e
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in step:
        a = agg[short(n)]
        a[0] += 1
        a[1] += e - s
    f.write("\nper kernel, this step:\n")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write("%8.1f us %5.1f%%  x%-3d %s\n" % (t / 1e3, 100.0 * t / busy, c, n))
print(open(out + "/timeline.txt").read().split("per kernel")[1][:3500])
PY
rm -rf $out/trace
