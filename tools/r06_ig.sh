#!/bin/bash
# round-6 development: one-barrier implicit GEMM with swapped MFMA operands (8-byte staging writes) -- variant tests, per-shape wall times against the round-5 library
mkdir -p gpurun_out; rm -f gpurun_out/r06_igemm_swap.txt
python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_san.py -q -x --tb=short 2>&1 | tail -3
SH="128,128,3,96,320 64,128,3,96,320 192,128,3,96,320 128,192,3,96,320 32,128,7,192,640 512,256,3,24,80 256,128,3,48,160 512,768,3,24,80 128,128,1,96,320 96,64,3,192,640 512,512,3,12,40"
for rep in 1 2; do
for lib in ab_lib/libmte_hip_r05.so ""; do
  echo "--- ${lib:-this tree}" >> gpurun_out/r06_igemm_swap.txt
  if [ -n "$lib" ]; then export MTE_LIB_PATH=$PWD/$lib; else unset MTE_LIB_PATH; fi
  python tools/conv_shape_bench.py $SH 2>/dev/null | cut -c1-92 >> gpurun_out/r06_igemm_swap.txt
done
done
unset MTE_LIB_PATH
cat gpurun_out/r06_igemm_swap.txt
