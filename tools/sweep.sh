#!/bin/bash
# development aid: bench under several MTE_DEBUG_KNOBS settings (one per argument)
mkdir -p gpurun_out; : > gpurun_out/sweep.txt
for k in "$@"; do
  [ "$k" = "base" ] && k=""
  echo "== knobs: $k" >> gpurun_out/sweep.txt
  MTE_DEBUG_KNOBS="$k" python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/sweep.txt
done
cat gpurun_out/sweep.txt
