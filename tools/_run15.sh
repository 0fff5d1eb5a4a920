python -m pytest tests/test_gpu_conv_variants.py -q -x --tb=line -k "eight_phase" > gpurun_out/r05_i8_tests.txt 2>&1; tail -3 gpurun_out/r05_i8_tests.txt
bash tools/ab_lib.sh $PWD/ab_lib/libmte_hip_tapmajor.so 3 2>&1 | tail -14
