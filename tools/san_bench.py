"""development aid: inference forward with and without the sparse (SAN) branch at 384x1280, bf16.  usage: san_bench.py [B]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mindtheedge_amd  # noqa: F401,E402  (sets GPU_MAX_HW_QUEUES before HIP starts)
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402
from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K.set_compute_dtype("bf16")
torch.manual_seed(0)
net = PackNetSAN01(dropout=None, version="1A", with_san=True).cuda().eval()
g = torch.Generator().manual_seed(1)
rgb = torch.rand(B, 3, 384, 1280, generator=g).cuda()
lidar = ((torch.rand(B, 1, 384, 1280, generator=g) < 0.05).float() * (2 + 70 * torch.rand(B, 1, 384, 1280, generator=g))).cuda()


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


with torch.no_grad():
    rgb_ms = timed(lambda: net(rgb))
    san_ms = timed(lambda: net(rgb, input_depth=lidar))
print(json.dumps({"B": B, "rgb_only_ms": round(rgb_ms, 3), "rgb_plus_lidar_ms": round(san_ms, 3), "san_branch_ms": round(san_ms - rgb_ms, 3),
                  "images_per_s_with_lidar": round(B / (san_ms * 1e-3), 1), "san_params": sum(p.numel() for p in net.mconvs.parameters())}))
