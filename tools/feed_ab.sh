#!/bin/bash
# A/B of the bf16 chains' weight feed (per-wave register ring vs the workgroup's LDS ring):
# the bf16 parity tests on the default feed, then the bf16 workloads on both feeds.
# Usage: bash tools/feed_ab.sh <tag>
set -e
out=gpurun_out/$1
mkdir -p $out
timeout -k 10 400 python -u -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py tests/test_gpu_bayes.py -x -q --timeout 200 --timeout-method thread > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for feed in ring coop; do
  for wl in "v1" "bayes" "v2"; do
    FPN_BF16_FEED=$feed timeout -k 10 200 python3 bench.py --config $wl --dtype bf16 --no-cpu-baseline --no-pointset --no-side-workloads > $out/${wl}_$feed.json 2> $out/${wl}_$feed.err
  done
done
python3 - $out <<'PY'
import json, sys
out = sys.argv[1]
for wl in ("v1", "bayes", "v2"):
    for feed in ("ring", "coop"):
        d = json.loads(open("%s/%s_%s.json" % (out, wl, feed)).read().strip().splitlines()[-1])
        k = d.get("kernels", {})
        print(wl, feed, round(d["value"]), "frustums/s;", ", ".join("%s %.3f" % (n, v["ms_fastest_step"] / (v["launches"] / 3)) for n, v in k.items() if "bf16" in n))
PY
