#!/bin/bash
# round-6 development: pack backward data on the matrix cores -- variant + layer tests, per-shape times (411 = fp32-VALU LDS stencil, 539 = MFMA [product], 555 = MFMA, one bf16 weight), step A/B
mkdir -p gpurun_out; out=gpurun_out/r06_pack_bwd_mfma.txt; rm -f $out
python -m pytest tests/test_gpu_pack3d_variants.py tests/test_gpu_layers.py tests/test_gpu_pack_fold.py -q -x --tb=short 2>&1 | tail -5 > $out
P3_ONLY=pack_bwd python tools/conv3d_bench.py 411 539 555 2>/dev/null >> $out
for rep in 1 2 3; do for k in 1=411 1=539; do
  MTE_USE_DEV_LIB=1 MTE_DEBUG_KNOBS=$k python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; h=d['roofline_hbm']
print('knob $k  %.2f img/s  %.2f ms/step  conv %.2f ms (frac %.3f)  gn %.2f ms' % (d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], h['ms_per_step']))" >> $out
done; done
cat $out
