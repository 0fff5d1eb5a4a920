#!/bin/bash
# development aid: bench with / without an environment switch.  usage: ab.sh VAR [bench args]
v=$1; shift
mkdir -p gpurun_out; : > gpurun_out/ab.txt
for rep in 1 2; do
  echo "== default" >> gpurun_out/ab.txt
  python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-kernel-timing "$@" 2>&1 | tail -1 | cut -c1-120 >> gpurun_out/ab.txt
  echo "== $v=1" >> gpurun_out/ab.txt
  export $v=1
  python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-kernel-timing "$@" 2>&1 | tail -1 | cut -c1-120 >> gpurun_out/ab.txt
  unset $v
done
cat gpurun_out/ab.txt
