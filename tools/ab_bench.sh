set -e
A=$PWD/3d-object-detection_amd/fpn/libfpn_A.so
for i in 1 2 3; do
  FPN_LIB_PATH=$A timeout -k 10 120 python -u bench.py --dtype bf16 --no-pointset --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/ab_A$i.json 2>/dev/null
  timeout -k 10 120 python -u bench.py --dtype bf16 --no-pointset --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/ab_B$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for v in "AB":
    for i in (1,2,3):
        d=json.loads(open("gpurun_out/ab_%s%d.json"%(v,i)).read().strip().splitlines()[-1])
        k=d["kernels"]
        print(v,i,"value %.0f"%d["value"], " ".join("%s=%.4f"%(n.replace("v1_","")[:22],x["ms_fastest_step"]) for n,x in k.items() if x["ms_fastest_step"]>0.03))
PY
