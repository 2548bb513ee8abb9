for g in 0 256 512; do
FPN_BF16_HEAD_GRID=$g timeout -k 10 120 python -u bench.py --no-pointset --no-cpu-baseline --steps 30 --warmup 5 --dtype bf16 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('grid $g value %.0f ' % d['value'] + ' '.join('%s=%.4f' % (n.replace('v1_','')[:22], x['ms_fastest_step']) for n,x in k.items() if x['ms_fastest_step']>0.03))
" || exit 1
done
