export MTE_BENCH_SHARE_GPU=1 PYTHONUNBUFFERED=1 PYTHONFAULTHANDLER=1
timeout -s INT 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --batch 1 --height 128 --width 256 --no-cpu-baseline --no-kernel-timing > gpurun_out/r2_share_full.txt 2>&1
grep -v amdgpu.ids gpurun_out/r2_share_full.txt | grep -v "^\s*$" | head -60 | cut -c1-400
