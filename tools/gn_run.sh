run() { python bench.py --steps 10 --warmup 4 --no-cpu-baseline --dump-conv gpurun_out/ab_$1.txt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', round(d['value'],1), round(d['ms_per_step'],2), 'conv', round(r['conv_ms_per_step'],2), round(r['frac'],4), 'ovl', round(r['conv_ms_per_step_overlapped'],2), {k:round(v['ms_per_step'],2) for k,v in r['by_kernel'].items()})"; }
python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
run tight
MTE_DEBUG_KNOBS=18=0 run loose
MTE_DEBUG_KNOBS=8=0 run tight_w128
MTE_DEBUG_KNOBS=18=0,8=0 run loose_w128
run tightb
MTE_DEBUG_KNOBS=18=0 run looseb
MTE_DEBUG_KNOBS=8=0 run tight_w128b
