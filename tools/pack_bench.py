#!/usr/bin/env python3
"""development aid (round 4): GPU time of the per-step weight re-pack (kernels.prefetch_weight_packs: the multi-tensor pack launches after the optimizer step),
timed alone on the current stream.  A/B two builds with MTE_LIB_PATH."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402
from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01  # noqa: E402
from mindtheedge_amd.models.SemiSupEdgeModel import SemiSupEdgeModel  # noqa: E402
from mindtheedge_amd.losses.grad_loss import GradLoss  # noqa: E402
from mindtheedge_amd.trainers.data_parallel import FlatParameters, FusedAdam  # noqa: E402

dev = torch.device("cuda", 0)
K.set_compute_dtype("bf16")
torch.manual_seed(42)
net = PackNetSAN01(dropout=0.5, version="1A").to(dev)
model = SemiSupEdgeModel(supervised_loss_weight=1.0, depth_edges_loss_weight=1.0, supervised_method="sparse-silog",
                         supervised_num_scales=1, edges_depth_edge_loss_all_scales=True, flip_lr_prob=0.5)
model.add_depth_net(net)
model.add_edge_loss(GradLoss("cross_entropy", True, [], 10.0, 1.0))
batch = bench.device_batch(2, 384, 1280, seed=1234, device=dev)
random.seed(100)
model.train()
flat = FlatParameters(net.parameters())
opt = FusedAdam(flat, lr=1e-4, reducer=None)
for _ in range(2):
    opt.zero_grad()
    model(batch)["loss"].backward()
    opt.step()
K.join_side_stream()
torch.cuda.synchronize()
K.use_wgrad_side_stream(False)
best = 1e9
for rep in range(10):
    K.bump_weights_epoch() if hasattr(K, "bump_weights_epoch") else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.prefetch_weight_packs()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3)
print("weight re-pack of the network: %.1f us" % best)
