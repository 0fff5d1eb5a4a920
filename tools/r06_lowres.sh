#!/bin/bash
# round-6 development: the 12x40 / 24x80 512-channel layers (few tiles, split K): tile forms by development knob -- 23 = 8-phase kernels (51 = product), 6 = older tile forms
mkdir -p gpurun_out; out=gpurun_out/r06_lowres_forms.txt; rm -f $out
SH="512,512,3,12,40 512,512,3,24,80 512,256,3,24,80 256,512,3,24,80"
for k in "" "23=0" "23=0,6=2" "23=0,6=1" "23=0,6=0"; do
  echo "--- knobs ${k:-product}" >> $out
  MTE_USE_DEV_LIB=1 MTE_DEBUG_KNOBS=$k python tools/conv_shape_bench.py $SH 2>/dev/null | cut -c1-110 >> $out
done
cat $out
