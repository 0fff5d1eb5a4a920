#!/bin/bash
# same-box A/B of two source trees (development aid): ab_old/ holds an earlier commit (git archive <rev> | tar -x -C ab_old; build it),
# the repository root the current one.  Alternates the two bench runs N times; prints value / ms_per_step / conv + GN family times.
n=${1:-2}; shift
mkdir -p gpurun_out
for i in $(seq 1 $n); do
  for t in ab_old .; do
    (cd $t && python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1) > gpurun_out/ab_$i.json
    python - "$t" gpurun_out/ab_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r, h = d.get("roofline", {}), d.get("roofline_hbm", {})
print("%-7s %7.2f img/s  %6.2f ms/step  conv %6.2f ms (frac %.3f)  gn %5.2f ms  host %5.2f ms" % (sys.argv[1], d["value"], d["ms_per_step"],
      r.get("conv_ms_per_step", 0), r.get("frac", 0), h.get("ms_per_step", 0), d.get("host_enqueue_unthrottled_ms_per_step", 0)))
PY
  done
done
