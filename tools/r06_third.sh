#!/bin/bash
# round-6 development call 3: T8 pin numbers, the inference error-word test, host operator census + memcpy sites of a step
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py -q -x -k "benchmark_batch_of_8" --timeout 900 -s 2>&1 | tail -5
cat gpurun_out/t8_pin.txt
python tools/torch_op_census.py > gpurun_out/r06_op_census.txt 2>&1; tail -150 gpurun_out/r06_op_census.txt
python tools/memcpy_sites.py > gpurun_out/r06_memcpy_sites.txt 2>&1; tail -80 gpurun_out/r06_memcpy_sites.txt
