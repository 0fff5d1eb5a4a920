#!/bin/bash
# same-box A/B of one environment switch of the host side (development aid): ab_env.sh VAR [reps]   e.g. ab_env.sh MTE_GN_IN_CONV 2
v=$1; n=${2:-2}
for i in $(seq 1 $n); do for f in 0 1; do
  env $v=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; h=d['roofline_hbm']; bk=r['by_kernel']
pf=sum(v['ms_per_step'] for k,v in bk.items() if 'patch_fwd' in k)
print('$v=$f  %.2f img/s  %.2f ms/step  conv %.2f ms (frac %.3f; patch fwd %.3f ms)  gn %.2f ms (%d launches)  untimed %s' % (d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], pf, h['ms_per_step'], h['launches_per_step'], r.get('untimed_conv_entry_points')))"
done; done
