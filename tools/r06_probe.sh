#!/bin/bash
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/read_bw tools/probe/read_bw.hip && /tmp/read_bw > gpurun_out/r06_read_bw.txt 2>&1; cat gpurun_out/r06_read_bw.txt
