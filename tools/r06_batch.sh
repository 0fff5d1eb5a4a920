#!/bin/bash
# round-6 probe: throughput against frames per step (is a smaller working set -- more of a layer's tensors in the 256 MB Infinity Cache between producer and consumer -- worth anything?)
mkdir -p gpurun_out; rm -f gpurun_out/r06_batch_sweep.txt
for b in 8 4 2 8 4 12 16; do
  python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('B=$b  %.2f img/s  %.2f ms/step  %.3f ms/image  host %.2f ms/step' % (d['value'], d['ms_per_step'], d['ms_per_step']/$b, d['host_enqueue_unthrottled_ms_per_step']))" >> gpurun_out/r06_batch_sweep.txt
done
cat gpurun_out/r06_batch_sweep.txt
