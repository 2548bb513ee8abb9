#!/bin/bash
# bf16 chain kernel times of a workload under each timing-only library variant in tools/bin
# (tools/build_variant.sh).  Usage: bash tools/variant_times.sh <tag> <bench flags...>
out=gpurun_out/$1; shift
mkdir -p $out
for lib in "" $(ls tools/bin/libfpn_*.so); do
  name=$(basename "${lib:-default}" .so)
  FPN_LIB_PATH=$lib timeout -k 10 120 python3 bench.py "$@" --no-cpu-baseline --no-pointset --no-side-workloads > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; exit 1; }
  python3 - $out/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.get("kernels", {})
print("%-14s %8.0f frustums/s; " % (sys.argv[2], d["value"]) + ", ".join("%s %.3f" % (n.replace("v1_", "").replace("_bf16", ""), v["ms_fastest_step"] / (v["launches"] / 3)) for n, v in k.items() if "bf16" in n))
PY
done
