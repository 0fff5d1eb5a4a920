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


from mindtheedge_amd.networks.layers import minkowski_encoder as me  # noqa: E402

with torch.no_grad():
    rgb_ms = timed(lambda: net(rgb))
    out = {"B": B, "rgb_only_ms": round(rgb_ms, 3), "san_params": sum(p.numel() for p in net.mconvs.parameters())}
    for gather in (False, True):                 # dense-equivalent convolutions of the zero-filled maps | gather-GEMM-scatter over the site lists
        me.SPARSE_GATHER = gather
        san_ms = timed(lambda: net(rgb, input_depth=lidar))
        tag = "gather" if gather else "dense"
        out.update({"rgb_plus_lidar_ms_" + tag: round(san_ms, 3), "san_branch_ms_" + tag: round(san_ms - rgb_ms, 3),
                    "images_per_s_with_lidar_" + tag: round(B / (san_ms * 1e-3), 1)})
    # active share per pyramid level (what the gather form's work scales with)
    net.mconvs.prep(lidar)
    dens = []
    for _ in range(5):
        net.mconvs()
        dens.append(round(float(net.mconvs.d[1].float().mean()), 3))
    out["active_share_per_level"] = dens
print(json.dumps(out))
