#!/bin/bash
# round-6 development call 2: M16 correctness (variant tests), per-shape A/B of the MFMA shape in the 5x5 / 7x7 second form, in-loop clocks with the buffer cleared
mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv_variants.py -q -x -k "patch_kernels_match or groupnorm_statistics" --timeout 900 > gpurun_out/r06_m16_tests.txt 2>&1; tail -5 gpurun_out/r06_m16_tests.txt
SH="32,32,7,384,1280 128,32,7,192,640 256,64,5,96,320 32,32,5,384,1280"
for k in 500 501 500 501; do
  echo "--- knob 11=$k" >> gpurun_out/r06_m16_ab.txt
  MTE_USE_DEV_LIB=1 MTE_DEBUG_KNOBS=11=$k python tools/conv_shape_bench.py $SH >> gpurun_out/r06_m16_ab.txt 2>&1
done
cat gpurun_out/r06_m16_ab.txt
python tools/inloop_clock.py 2.0 > gpurun_out/r06_inloop_clock.txt 2>gpurun_out/r06_inloop_clock.err
cat gpurun_out/r06_inloop_clock.txt
