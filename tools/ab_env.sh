# A/B of one environment switch on a bench.py workload: alternating runs, per-kernel fastest-step times.
# Usage: bash tools/ab_env.sh VAR valueA valueB [bench.py flags]
set -e
var=$1; a=$2; b=$3; shift 3
for i in 1 2 3; do
  for v in $a $b; do
    env $var=$v timeout -k 10 120 python -u bench.py --no-pointset --no-cpu-baseline --steps 30 --warmup 5 "$@" > gpurun_out/ab_${v}_$i.json 2>/dev/null
  done
done
python - $a $b <<'PY'
import json, sys
for v in sys.argv[1:3]:
    for i in (1, 2, 3):
        d = json.loads(open("gpurun_out/ab_%s_%d.json" % (v, i)).read().strip().splitlines()[-1])
        k = d["kernels"]
        print("%-8s %d value %.0f " % (v, i, d["value"]) + " ".join("%s=%.4f" % (n.replace("v1_", "")[:22], x["ms_fastest_step"]) for n, x in k.items() if x["ms_fastest_step"] > 0.03))
PY
