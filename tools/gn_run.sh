python tools/host_profile.py 2>&1 | grep "host ms"
python -m pytest tests/test_gpu_layers.py tests/test_gpu_network.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
