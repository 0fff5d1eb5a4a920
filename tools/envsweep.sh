#!/bin/bash
# development aid: bench under several values of one environment variable.  usage: envsweep.sh VAR v1 v2 ...
v=$1; shift
mkdir -p gpurun_out; : > gpurun_out/envsweep.txt
for val in "$@"; do
  export $v=$val
  echo "== $v=$val" >> gpurun_out/envsweep.txt
  python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])" >> gpurun_out/envsweep.txt
done
cat gpurun_out/envsweep.txt
