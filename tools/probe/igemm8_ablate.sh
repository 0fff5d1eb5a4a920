#!/bin/bash
# development aid: same-box timing of conv_igemm8.hip variants (stamped builds): the
# current source with pieces left out (MTE8_ABL bit 0: tap evaluation, bit 1: tap stepping, bit 2: s_setprio around the MFMAs)
mkdir -p gpurun_out
S="256,256,3,48,160 128,128,3,96,320 32,128,7,192,640"
{
for abl in ${ABLS:-0 4 7}; do IGEMM8_DEFS="-DMTE8_ABL=$abl" python tools/igemm8_stamps.py $S; done
} > gpurun_out/igemm8_ablate.log 2>&1
grep -v "amdgpu.ids\|group 1" gpurun_out/igemm8_ablate.log
