#!/bin/bash
# round-6 development: folded pack layers' data gradient without the pixel-shuffle pass (mte_conv2d_igemm_unshuffle) -- tests, step A/B (MTE_NO_UNSHUFFLE=1: the two-launch path)
mkdir -p gpurun_out; out=gpurun_out/r06_unshuffle.txt; rm -f $out
python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_pack_fold.py tests/test_gpu_layers.py tests/test_gpu_fullsize.py -q -x --tb=short 2>&1 | tail -5 >> $out
for rep in 1 2 3; do for p in 1 0; do
  MTE_NO_UNSHUFFLE=$p python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; h=d['roofline_hbm']
print('MTE_NO_UNSHUFFLE=$p  %.2f img/s  %.2f ms/step  conv %.2f ms (frac %.3f, untimed %s)  gn %.2f ms' % (d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], r.get('untimed_conv_entry_points'), h['ms_per_step']))" >> $out
done; done
cat $out
