#!/bin/bash
# development aid: same-box A/B of two builds of the library WITH the per-family kernel timing.  usage: ab_lib2.sh <other.so> [reps]
other=$1; reps=${2:-2}
mkdir -p gpurun_out
for rep in $(seq $reps); do
  for lib in "" "$other"; do
    export MTE_LIB_PATH=$lib
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_lib_$rep.json
    python - "${lib:-in-tree}" gpurun_out/ab_lib_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r, h = d.get("roofline", {}), d.get("roofline_hbm", {})
bk = r.get("by_kernel", {})
print("%-22s %7.2f img/s %6.2f ms/step  conv %6.2f ms (frac %.3f)  igemm %5.2f ms" % (sys.argv[1][-22:], d["value"], d["ms_per_step"],
      r.get("conv_ms_per_step", 0), r.get("frac", 0), bk.get("mte_conv2d_igemm", {}).get("ms_per_step", 0)))
PY
  done
done
