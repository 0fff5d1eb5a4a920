"""development aid: time the on-device annotation post-processing (row f-2) on a 4-scale pyramid at 384x1280 and the numpy
oracle beside it.  usage: python tools/dee_bench.py [B] [--cpu]  ->  one JSON line"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindtheedge_amd.utils import tools                                     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4


def edge_map(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    m = torch.zeros(B, H, W)
    for b in range(B):
        for _ in range(40):
            cx, cy, rad = torch.rand(3, generator=g) * torch.tensor([W, H, 60.0])
            d = (torch.sqrt((x - cx) ** 2 + (y - cy) ** 2) - (8 + rad)).abs()
            m[b] = torch.maximum(m[b], torch.exp(-0.5 * (d / 1.3) ** 2) * (0.3 + 0.7 * torch.rand(1, generator=g)))
    return (m * (0.5 + 0.5 * torch.rand(B, H, W, generator=g)) + 0.2 * torch.rand(B, H, W, generator=g) ** 4).clamp(0, 1)


preds_cpu = [(edge_map(B, 384 >> s, 1280 >> s, seed=s) * 2.0).unsqueeze(1) for s in range(4)]
preds = [p.cuda() for p in preds_cpu]
for _ in range(3):
    tools.annotate_edges(preds)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    out = tools.annotate_edges(preds)
torch.cuda.synchronize()
wall_ms = (time.perf_counter() - t0) * 1e3 / N
px = sum(B * (384 >> s) * (1280 >> s) for s in range(4))
res = {"B": B, "scales": 4, "pixels_per_batch": px, "wall_ms_per_batch": round(wall_ms, 3), "maps_per_s": round(4 * B / (wall_ms * 1e-3), 1),
       "images_per_s": round(B / (wall_ms * 1e-3), 1)}
if "--cpu" in sys.argv:
    from oracle import dee_oracle as do
    t0 = time.perf_counter()
    for s in range(4):
        do.annotate((preds_cpu[s][0, 0].numpy() / 2).astype(np.float32))
    res["cpu_oracle_ms_per_image"] = round((time.perf_counter() - t0) * 1e3, 1)
print(json.dumps(res))
