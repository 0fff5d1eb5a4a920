#!/bin/bash
# development aid (round 5): the nine-tap 3x3 weight gradient (conv_wgrad9.hip) against the per-tap kernel at the T8 shapes it takes over.
# [1] the variant test, [2] wall time incl. the unpack pass for knob 26 = 0 / 1, [3] kernel-only times from rocprofv3 for both.
mkdir -p gpurun_out
python -m pytest tests/test_gpu_conv_variants.py -q -x --tb=short -k "nine_tap or dma_wgrad" > gpurun_out/r05_wgrad9_test.txt 2>&1; tail -15 gpurun_out/r05_wgrad9_test.txt
SH="512,512,3,24,80 256,512,3,24,80 512,256,3,24,80 768,512,3,24,80 4096,256,3,24,80 256,256,3,48,160 128,256,3,48,160 384,256,3,48,160 128,128,3,96,320"
export MTE_USE_DEV_LIB=1
for k in 0 1 2 1 2; do
  echo "== knob 26 = $k" >> gpurun_out/r05_wgrad9_bench.txt
  MTE_DEBUG_KNOBS=26=$k python tools/conv_shape_bench.py $SH 2>&1 | sed -e 's/.*wgrad/wgrad/' >> gpurun_out/r05_wgrad9_bench.txt
done
cat gpurun_out/r05_wgrad9_bench.txt
cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
  rm -rf /tmp/w9p$k
  MTE_DEBUG_KNOBS=26=$k rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/w9p$k -- python3 $GRAFT_REPO_ROOT/tools/conv_shape_bench.py $SH > /dev/null 2>&1
  f=$(find /tmp/w9p$k -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats, knob 26 = $k" >> $GRAFT_REPO_ROOT/gpurun_out/r05_wgrad9_kernels.txt
  grep -i "wgrad\|unpack\|reduce_parts" "$f" | cut -c1-220 >> $GRAFT_REPO_ROOT/gpurun_out/r05_wgrad9_kernels.txt
done
cat $GRAFT_REPO_ROOT/gpurun_out/r05_wgrad9_kernels.txt
