python -m pytest tests/ -q -m gpu 2>&1 | tail -12
echo "=== bench default"; python bench.py > gpurun_out/r2_b3.json 2> gpurun_out/r2_b3.err; tail -c 3000 gpurun_out/r2_b3.json; tail -3 gpurun_out/r2_b3.err
