#!/bin/bash
# development aid: GroupNorm launch geometry (knobs 2 = min rows per thread row, 3 = target workgroups) vs step time and the family's serial time
mkdir -p gpurun_out; : > gpurun_out/gn_geom.txt
for k in "" "2=16" "2=8" "2=8,3=4096" "2=16,3=4096" ""; do
  echo "== knobs: $k" >> gpurun_out/gn_geom.txt
  MTE_DEBUG_KNOBS="$k" python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); h=d['roofline_hbm']; print(round(d['value'],1), round(d['ms_per_step'],3), 'gn ms', round(h['ms_per_step'],3), 'frac', round(h['frac'],3))" >> gpurun_out/gn_geom.txt
done
cat gpurun_out/gn_geom.txt
