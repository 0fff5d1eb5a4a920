#!/bin/bash
# Runs the GPU parity suite without -x and keeps the full report under gpurun_out/ (development aid).
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short --timeout 600 "$@" > gpurun_out/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.txt
tail -60 gpurun_out/pytest_gpu.txt
