#!/bin/bash
# round-6 development call 5: read-only stream ceiling (probe), T8 pin with the block measure, inference error-word test
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/read_bw tools/probe/read_bw.hip && /tmp/read_bw > gpurun_out/r06_read_bw.txt 2>&1; cat gpurun_out/r06_read_bw.txt
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_entry_points.py -q -x --timeout 900 -s 2>&1 | tail -8
cat gpurun_out/t8_pin.txt
