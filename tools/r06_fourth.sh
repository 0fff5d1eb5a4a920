#!/bin/bash
# round-6 development call 4: M16 everywhere in the LDS-patch forward -- parity tests, per-shape A/B by knob, step A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_layers.py tests/test_gpu_network.py -q -x --timeout 900 > gpurun_out/r06_m16_tests2.txt 2>&1; tail -5 gpurun_out/r06_m16_tests2.txt
SH="64,64,3,192,640 32,64,3,192,640 64,32,3,192,640 72,32,3,384,1280,96 32,64,3,384,1280 128,64,3,96,320 104,64,3,192,640,128 64,64,1,192,640 32,32,3,384,1280"
rm -f gpurun_out/r06_m16_ab2.txt
for rep in 1 2; do
for k in "11=600,11=700" "11=601,11=701"; do
  echo "--- knobs $k" >> gpurun_out/r06_m16_ab2.txt
  MTE_USE_DEV_LIB=1 MTE_DEBUG_KNOBS=$k python tools/conv_shape_bench.py $SH >> gpurun_out/r06_m16_ab2.txt 2>&1
done
done
cat gpurun_out/r06_m16_ab2.txt
bash tools/ab_trees.sh 2 > gpurun_out/r06_ab_trees_2.txt 2>&1
cat gpurun_out/r06_ab_trees_2.txt
