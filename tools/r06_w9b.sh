#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_groupnorm.py tests/test_gpu_layers.py -q -x --timeout 900 2>&1 | tail -3
bash tools/ab_trees.sh 3 > gpurun_out/r06_ab_trees_3.txt 2>&1; cat gpurun_out/r06_ab_trees_3.txt
