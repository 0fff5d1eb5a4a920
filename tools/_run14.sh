python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_determinism.py tests/test_gpu_layers.py tests/test_gpu_network.py -q -x --tb=short > gpurun_out/r05_i8_tests.txt 2>&1; tail -6 gpurun_out/r05_i8_tests.txt
python tools/igemm_race_stress.py 10 1 2>&1 | tail -12
python tools/igemm8_check.py bench 2>&1 | tail -32
