# One line per run: whole-job rate and the fastest-step time of every kernel above 30 us.
# Usage: bash tools/quick_bench.sh [bench.py flags]   (e.g. --dtype bf16)
timeout -k 10 200 python -u bench.py --no-pointset --no-cpu-baseline --no-side-workloads --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('value %.0f frac %.3f ' % (d['value'], d['roofline']['frac']) + ' '.join('%s=%.4f' % (n.replace('v1_','')[:22], x['ms_fastest_step']) for n,x in k.items() if x['ms_fastest_step']>0.03))
"
