#!/bin/bash
# development aid: marginal cost of entry points -- bench with each of them turned into a no-op (results are garbage, timing only)
mkdir -p gpurun_out; : > gpurun_out/skipsweep.txt
for k in "" "$@"; do
  export MTE_SKIP=$k
  echo "== skip: ${k:-nothing}" >> gpurun_out/skipsweep.txt
  python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/skipsweep.txt 2>&1
done
cat gpurun_out/skipsweep.txt
