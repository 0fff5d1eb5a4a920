#!/bin/bash
# round-6 development: weight-fragment ring refills pinned (first form + 5x5 / 7x7 second form) -- variant tests, per-shape A/B against the round-5 library, step A/B
mkdir -p gpurun_out; rm -f gpurun_out/r06_ring_refills.txt
python -m pytest tests/test_gpu_conv_variants.py -q -x --tb=short -k "patch" 2>&1 | tail -3
SH="32,32,7,384,1280 128,32,7,192,640 256,64,5,96,320 32,32,5,384,1280 64,64,3,192,640 32,64,3,192,640 64,32,3,192,640 64,64,1,192,640"
for rep in 1 2; do
for lib in ab_lib/libmte_hip_r05.so ""; do
  echo "--- ${lib:-this tree}" >> gpurun_out/r06_ring_refills.txt
  if [ -n "$lib" ]; then export MTE_LIB_PATH=$PWD/$lib; else unset MTE_LIB_PATH; fi
  python tools/conv_shape_bench.py $SH 2>/dev/null | cut -c1-118 >> gpurun_out/r06_ring_refills.txt
done
done
unset MTE_LIB_PATH
cat gpurun_out/r06_ring_refills.txt
bash tools/ab_trees.sh 2 > gpurun_out/r06_ab_trees_4.txt 2>&1; cat gpurun_out/r06_ab_trees_4.txt
