python -m pytest tests/test_gpu_san.py tests/test_gpu_entry_points.py -x -q -m gpu 2>&1 | tail -30
