#!/bin/bash
# round-6 development: weight gradients forked BEHIND the data-gradient launch (MTE_WGRAD_AFTER_DGRAD=1) -- same box, alternating
mkdir -p gpurun_out; out=gpurun_out/r06_wgrad_after_dgrad.txt; rm -f $out
for rep in 1 2 3; do for p in 0 1; do
  MTE_WGRAD_AFTER_DGRAD=$p python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; h=d['roofline_hbm']
print('MTE_WGRAD_AFTER_DGRAD=$p  %.2f img/s  %.2f ms/step  conv %.2f ms (frac %.3f)  gn %.2f ms' % (d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], h['ms_per_step']))" >> $out
done; done
cat $out
