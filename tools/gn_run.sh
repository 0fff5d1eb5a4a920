export MTE_BENCH_DIST_SELFTEST=1 PYTHONUNBUFFERED=1
timeout -s INT 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-2500
