#!/usr/bin/env python3
"""development aid (round 4): how far ahead of the GPU the host runs at the phase boundaries of a training step.  At each boundary the host
notes its clock and records an event on the main stream; lead = (time the GPU reaches the event) - (time the host recorded it).  A lead near
zero means the GPU had caught up with the host there: whatever the host does next (autograd engine start-up, optimizer, python between the
phases) is GPU idle time.  usage: host_lead.py [steps]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402
from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01  # noqa: E402
from mindtheedge_amd.models.SemiSupEdgeModel import SemiSupEdgeModel  # noqa: E402
from mindtheedge_amd.losses.grad_loss import GradLoss  # noqa: E402
from mindtheedge_amd.trainers.data_parallel import FlatParameters, FusedAdam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
K.set_compute_dtype("bf16")
torch.manual_seed(42)
net = PackNetSAN01(dropout=0.5, version="1A").to(dev)
model = SemiSupEdgeModel(supervised_loss_weight=1.0, depth_edges_loss_weight=1.0, supervised_method="sparse-silog",
                         supervised_num_scales=1, edges_depth_edge_loss_all_scales=True, flip_lr_prob=0.5)
model.add_depth_net(net)
model.add_edge_loss(GradLoss("cross_entropy", True, [], 10.0, 1.0))
batch = bench.device_batch(8, 384, 1280, seed=1234, device=dev)
random.seed(100)
model.train()
flat = FlatParameters(net.parameters())
opt = FusedAdam(flat, lr=1e-4, reducer=None)
marks = []


def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((name, time.perf_counter(), ev))


for s in range(steps):
    mark("step start")
    opt.zero_grad()
    mark("after zero_grad")
    out = model(batch)
    mark("after forward + loss")
    out["loss"].backward()
    mark("after backward")
    opt.step()
    mark("after optimizer")
torch.cuda.synchronize()
base_name, base_t, base_ev = marks[0]
# host clock and event clock tick at the same rate; anchor them at the LAST mark, where the host waited for the GPU (synchronize right after)
rows = [(n, (t - base_t) * 1e3, base_ev.elapsed_time(ev)) for n, t, ev in marks]
print("%-24s %10s %10s %10s" % ("boundary", "host ms", "gpu ms", "lead ms"))
off = None
for i, (n, th, tg) in enumerate(rows):
    if i < 5 * (steps - 4):
        continue
    print("%-24s %10.2f %10.2f %10.2f" % (n, th, tg, tg - th))
