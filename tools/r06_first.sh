#!/bin/bash
# round-6 development call 1: in-loop clock of the MFMA loops, phase stamps of the LDS-patch forward, the LDS-size fix per shape (this tree against the
# round-5 library in ab_lib/), same-box A/B of the round-5 tree (ab_old/) against this one.
mkdir -p gpurun_out
python tools/inloop_clock.py 2.0 > gpurun_out/r06_inloop_clock.txt 2>gpurun_out/r06_inloop_clock.err
tail -40 gpurun_out/r06_inloop_clock.txt
python tools/patch_stamps.py 32,32,7,384,1280 128,32,7,192,640 256,64,5,96,320 64,32,3,192,640 72,32,3,384,1280,96 > gpurun_out/r06_patch_stamps.txt 2>&1
tail -30 gpurun_out/r06_patch_stamps.txt
SH="64,32,3,192,640 72,32,3,384,1280,96 32,64,3,384,1280 32,32,7,384,1280 64,64,3,192,640 32,64,3,192,640"
echo "--- this tree" > gpurun_out/r06_lds_fix.txt
python tools/conv_shape_bench.py $SH >> gpurun_out/r06_lds_fix.txt 2>&1
echo "--- round-5 library" >> gpurun_out/r06_lds_fix.txt
MTE_LIB_PATH=$PWD/ab_lib/libmte_hip_r05.so python tools/conv_shape_bench.py $SH >> gpurun_out/r06_lds_fix.txt 2>&1
cat gpurun_out/r06_lds_fix.txt
bash tools/ab_trees.sh 2 > gpurun_out/r06_ab_trees_1.txt 2>&1
cat gpurun_out/r06_ab_trees_1.txt
