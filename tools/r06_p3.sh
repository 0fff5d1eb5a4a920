#!/bin/bash
# round-6 development: conv3d kernels with batched tile fills -- tests, conv3d_bench against the round-5 development library, p3 weight-gradient bench
mkdir -p gpurun_out; rm -f gpurun_out/r06_p3_fill.txt
python -m pytest tests/test_gpu_pack3d_variants.py tests/test_gpu_layers.py -q -x --tb=short -k "pack or unpack or conv3d" 2>&1 | tail -3
for rep in 1 2; do
for lib in ab_lib/libmte_hip_dev_r05.so ""; do
  echo "--- ${lib:-this tree}" >> gpurun_out/r06_p3_fill.txt
  if [ -n "$lib" ]; then export MTE_LIB_PATH=$PWD/$lib; else unset MTE_LIB_PATH; fi
  python tools/conv3d_bench.py 411 2>/dev/null | cut -c1-150 >> gpurun_out/r06_p3_fill.txt
  python tools/p3_wgrad_bench.py 2>/dev/null | cut -c1-150 >> gpurun_out/r06_p3_fill.txt
done
done
unset MTE_LIB_PATH
cat gpurun_out/r06_p3_fill.txt
