"""development aid: time the on-device validation metrics (row f-3) at KITTI size and the CPU oracle beside it.
usage: python tools/metrics_bench.py [B]   ->  one JSON line"""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindtheedge_amd.utils.depth import compute_depth_metrics, post_process_inv_depth      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H, W, h, w = 375, 1242, 384, 1280
g = torch.Generator().manual_seed(0)
gt = (0.5 + 95 * torch.rand(B, 1, H, W, generator=g) ** 2) * (torch.rand(B, 1, H, W, generator=g) > 0.8).float()
inv = 0.01 + torch.rand(B, 1, h, w, generator=g)
invf = 0.01 + torch.rand(B, 1, h, w, generator=g)
cfg = types.SimpleNamespace(crop="garg", scale_output="resize", min_depth=0.0, max_depth=80.0)
gt_d, inv_d, invf_d = gt.cuda(), inv.cuda(), invf.cuda()


def device_pass():
    pp = post_process_inv_depth(inv_d, invf_d, "mean")
    d, dpp = 1.0 / inv_d.clamp(min=1e-6), 1.0 / pp.clamp(min=1e-6)
    return [compute_depth_metrics(cfg, gt_d, dpp if "pp" in m else d, use_gt_scale="gt" in m) for m in ("", "_pp", "_gt", "_pp_gt")]


for _ in range(5):
    device_pass()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 50
e0.record()
for _ in range(N):
    device_pass()
e1.record()
torch.cuda.synchronize()
dev_ms = e0.elapsed_time(e1) / N
t0 = time.perf_counter()
for _ in range(N):
    out = device_pass()
torch.cuda.synchronize()
wall_ms = (time.perf_counter() - t0) * 1e3 / N

res = {"B": B, "gt": [H, W], "pred": [h, w], "device_ms_per_batch": round(dev_ms, 4), "wall_ms_per_batch": round(wall_ms, 4),
       "images_per_s": round(B / (wall_ms * 1e-3), 1)}
# algorithmic bytes: fusion 12 B/px of the prediction; metrics: (4 B gt + 4 B pred) per cropped pixel per pass, 1 pass without
# and 4 with median scaling, two calls each
crop_px = (int(0.99189189 * H) - int(0.40810811 * H)) * (int(0.96405229 * W) - int(0.03594771 * W))
res["algorithmic_MB_per_batch"] = round(B * (12 * h * w + 8 * crop_px * (1 + 1 + 4 + 4)) / 1e6, 2)
res["achieved_GBps"] = round(res["algorithmic_MB_per_batch"] / 1e3 / (dev_ms * 1e-3), 1)
if "--cpu" in sys.argv:
    from oracle import metrics_oracle as mo
    t0 = time.perf_counter()
    pp = mo.post_process_inv_depth(inv.numpy(), invf.numpy(), "mean")
    d, dpp = 1.0 / inv.clamp(min=1e-6).numpy(), 1.0 / pp.clip(min=1e-6)
    for m in ("", "_pp", "_gt", "_pp_gt"):
        mo.compute_depth_metrics(gt.numpy(), dpp if "pp" in m else d, crop="garg", use_gt_scale="gt" in m)
    res["cpu_oracle_ms_per_batch"] = round((time.perf_counter() - t0) * 1e3, 1)
print(json.dumps(res))
