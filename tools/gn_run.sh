run() { python bench.py --steps 10 --warmup 4 --no-cpu-baseline --dump-conv gpurun_out/ab_$1.txt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', round(d['value'],1), round(d['ms_per_step'],2), 'conv', round(r['conv_ms_per_step'],2), round(r['frac'],4), 'ovl', round(r['conv_ms_per_step_overlapped'],2), {k:round(v['ms_per_step'],2) for k,v in r['by_kernel'].items()})"; }
run mfma3d
MTE_DEBUG_KNOBS=1=200 run valu3d
run mfma3db
MTE_DEBUG_KNOBS=1=200 run valu3db
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
