python tools/graph_debug.py 2>&1 | grep -v amdgpu.ids | tail -3
python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8
