#!/bin/bash
# round-6 development: nine-tap weight gradient without the spilled lane flags -- variant test, then wall times against the round-5 development library
mkdir -p gpurun_out; rm -f gpurun_out/r06_wgrad9_nospill.txt
python -m pytest tests/test_gpu_conv_variants.py -q -x --tb=short -k "nine_tap or dma_wgrad or launch_width" 2>&1 | tail -3
SH="512,512,3,24,80 256,512,3,24,80 768,512,3,24,80 4096,256,3,24,80 256,256,3,48,160 384,256,3,48,160 128,128,3,96,320 192,128,3,96,320"
for rep in 1 2 3; do
for lib in ab_lib/libmte_hip_r05.so ""; do
  echo "--- ${lib:-this tree}" >> gpurun_out/r06_wgrad9_nospill.txt
  if [ -n "$lib" ]; then export MTE_LIB_PATH=$PWD/$lib; else unset MTE_LIB_PATH; fi
  python tools/conv_shape_bench.py $SH 2>/dev/null | sed -e 's/fwd.*wgrad/wgrad/' >> gpurun_out/r06_wgrad9_nospill.txt
done
done
cat gpurun_out/r06_wgrad9_nospill.txt
