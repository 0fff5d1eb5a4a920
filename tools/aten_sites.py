#!/usr/bin/env python3
"""development aid (round 6): every top-level ATen operator of ONE training step with its input shapes and the autograd node / python function it runs under
(the profiler's parent chain) -- who still launches element-wise kernels on the main queue.  usage: aten_sites.py"""
import os
import random
import sys
from collections import Counter

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
                         supervised_num_scales=1, edges_depth_edge_loss_all_scales=True, flip_lr_prob=0.0)
model.add_depth_net(net)
model.add_edge_loss(GradLoss("cross_entropy", True, [], 10.0, 1.0))
batch = bench.device_batch(8, 384, 1280, seed=1234, device=dev)
random.seed(100)
model.train()
flat = FlatParameters(net.parameters())
opt = FusedAdam(flat, lr=1e-4, reducer=None)


def step():
    opt.zero_grad()
    out = model(batch)
    out["loss"].backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = Counter()
for ev in prof.events():
    n = ev.name
    if not n.startswith("aten::") or n in ("aten::slice", "aten::as_strided", "aten::set_", "aten::select", "aten::view", "aten::empty", "aten::detach", "aten::alias",
                                           "aten::empty_like", "aten::empty_strided", "aten::permute", "aten::expand", "aten::lift_fresh", "aten::reshape", "aten::t",
                                           "aten::transpose", "aten::unsqueeze", "aten::squeeze", "aten::contiguous", "aten::_unsafe_view", "aten::result_type", "aten::item",
                                           "aten::_local_scalar_dense", "aten::is_nonzero", "aten::to", "aten::narrow", "aten::unbind", "aten::view_as", "aten::clone"):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    p, chain = ev.cpu_parent, []
    while p is not None and len(chain) < 3:
        chain.append(p.name[:70])
        p = p.cpu_parent
    shapes = str([s for s in (ev.input_shapes or []) if s])[:70]
    rows[(n, shapes, " < ".join(chain))] += 1
print("top-level ATen operators with kernels in ONE training step (no flip): %d" % sum(rows.values()))
for (n, shapes, chain), c in sorted(rows.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%3d  %-22s %-72s %s" % (c, n, shapes, chain))
