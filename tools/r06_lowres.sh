#!/bin/bash
# round-6 development: split-K 8-phase launches with very few 256 x 256 tiles (12x40 / 24x80 512-channel layers): 256 x 128 tiles with half the K splits (knob 29 = tile bound)
mkdir -p gpurun_out; out=gpurun_out/r06_lowres_forms.txt; rm -f $out
SH="512,512,3,12,40 512,512,3,24,80 512,256,3,24,80 256,512,3,24,80 8192,512,3,12,40 4096,256,3,24,80"
for rep in 1 2; do for k in "" "29=32" "29=64"; do
  echo "--- knobs ${k:-product}" >> $out
  MTE_USE_DEV_LIB=1 MTE_DEBUG_KNOBS=$k python tools/conv_shape_bench.py $SH 2>/dev/null | cut -c1-110 >> $out
done; done
cat $out
