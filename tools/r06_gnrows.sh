#!/bin/bash
# round-6 development: GroupNorm stream kernels' launch geometry on the low / mid resolution layers (knob 2 = min rows per thread row, knob 3 = target workgroups) -- step A/B, dev library
mkdir -p gpurun_out; out=gpurun_out/r06_gn_rows.txt; rm -f $out
for rep in 1 2; do for k in "" "2=48" "2=64" "2=24"; do
  MTE_USE_DEV_LIB=1 MTE_DEBUG_KNOBS=$k python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; h=d['roofline_hbm']
print('knobs %-14s %.2f img/s  %.2f ms/step  conv %.2f ms  gn %.3f ms (frac %.3f)' % ('${k:-default}', d['value'], d['ms_per_step'], r['conv_ms_per_step'], h['ms_per_step'], h['frac']))" >> $out
done; done
cat $out
