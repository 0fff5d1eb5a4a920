#!/bin/bash
out=gpurun_out/r06_repro_bisect3.txt; mkdir -p gpurun_out; rm -f $out
MTE_USE_DEV_LIB=1 python tools/_diag_repro.py 40 tree-dev 2>/dev/null | tail -1 >> $out
python tools/_diag_repro.py 40 tree-product 2>/dev/null | tail -1 >> $out
cat $out
bash tools/r06_ig.sh 2>&1 | tail -30
