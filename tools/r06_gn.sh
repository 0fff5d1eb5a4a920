#!/bin/bash
# round-6 development: GroupNorm backward reduce pass -- round-5 library, this tree (GN_RED_U = 4), private builds with 2 / 8 rows per batch; gn tests
mkdir -p gpurun_out
python -m pytest tests/test_gpu_groupnorm.py -q -x --timeout 900 2>&1 | tail -3
rm -f gpurun_out/r06_gn_reduce.txt
for rep in 1 2; do
for lib in ab_lib/libmte_hip_dev_r05.so "" ab_lib/libmte_hip_dev_u2.so ab_lib/libmte_hip_dev_u8.so; do
  echo "--- ${lib:-this tree (U = 4)}" >> gpurun_out/r06_gn_reduce.txt
  if [ -n "$lib" ]; then export MTE_LIB_PATH=$PWD/$lib; else unset MTE_LIB_PATH; fi
  python tools/gn_bench.py 2>/dev/null | grep -v "^$" >> gpurun_out/r06_gn_reduce.txt
done
done
cat gpurun_out/r06_gn_reduce.txt
