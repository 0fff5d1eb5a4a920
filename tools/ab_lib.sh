#!/bin/bash
# development aid: same-box A/B of two builds of the library.  usage: ab_lib.sh <other.so> [reps]
other=$1; reps=${2:-2}
mkdir -p gpurun_out; : > gpurun_out/ab_lib.txt
for rep in $(seq $reps); do
  for lib in "" "$other"; do
    export MTE_LIB_PATH=$lib
    echo "== lib: ${lib:-in-tree}" >> gpurun_out/ab_lib.txt
    python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/ab_lib.txt
  done
done
cat gpurun_out/ab_lib.txt
