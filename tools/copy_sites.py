#!/usr/bin/env python3
"""development aid (round 4): who issues the device-to-device copies of a training step (__amd_rocclr_copyBuffer in the kernel trace: ~100 per step,
3.7 us each)?  Counts Tensor.copy_ / clone / contiguous / float / to calls that really copy on the device during ONE step, by call site."""
import collections
import os
import random
import sys
import traceback

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


for _ in range(3):
    step()
torch.cuda.synchronize()
sites = collections.Counter()
active = [True]


def site():
    st = traceback.extract_stack(limit=8)[:-2]
    own = [f for f in st if "mindtheedge_amd" in f.filename or f.filename.endswith("bench.py")]
    f = own[-1] if own else st[-1]
    return "%s:%d %s" % (os.path.relpath(f.filename, ROOT), f.lineno, f.line)


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def w(self, *a, **k):
        r = orig(self, *a, **k)
        if active[0] and self.is_cuda:
            copied = name == "copy_" or name == "clone" or (isinstance(r, torch.Tensor) and r.data_ptr() != self.data_ptr())
            if copied:
                active[0] = False
                sites[(name, site(), tuple(self.shape) if self.dim() < 3 else self.dim())] += 1
                active[0] = True
        return r
    setattr(torch.Tensor, name, w)


for n in ("copy_", "clone", "contiguous", "float", "to", "zero_", "fill_"):
    wrap(n)
step()
torch.cuda.synchronize()
for (name, s, shp), c in sites.most_common(40):
    print("%4d  %-10s %s   %s" % (c, name, s[:150], shp))
