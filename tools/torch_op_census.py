#!/usr/bin/env python3
"""development aid: which torch (ATen) operators and memcpys a training step still issues besides the library's own launches --
every one is a host dispatch (~10-20 us) and a small kernel on the main chain.  usage: torch_op_census.py [B]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
K.set_compute_dtype("bf16")
torch.manual_seed(42)
net = PackNetSAN01(dropout=0.5, version="1A").to(dev)
model = SemiSupEdgeModel(supervised_loss_weight=1.0, depth_edges_loss_weight=1.0, supervised_method="sparse-silog",
                         supervised_num_scales=1, edges_depth_edge_loss_all_scales=True, flip_lr_prob=0.5)
model.add_depth_net(net)
model.add_edge_loss(GradLoss("cross_entropy", True, [], 10.0, 1.0))
batch = bench.device_batch(B, 384, 1280, seed=1234, device=dev)
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

# ---- call sites (python level): wrap the tensor methods / factory functions the step uses and record the caller
sites = Counter()


def _wrap(owner, name):
    orig = getattr(owner, name)

    def w(*a, **k):
        f = sys._getframe(1)
        depth = 0
        while f is not None and "mindtheedge_amd" not in f.f_code.co_filename and depth < 6:
            f, depth = f.f_back, depth + 1
        if f is not None:
            sites[(name, "%s:%d" % (f.f_code.co_filename.split("/root/repo/")[-1].split("mindtheedge_amd/")[-1], f.f_lineno))] += 1
        return orig(*a, **k)
    setattr(owner, name, w)
    return orig


saved = [(torch.Tensor, n, _wrap(torch.Tensor, n)) for n in ("copy_", "add_", "to", "float", "contiguous", "permute", "view", "zero_", "detach", "clone", "mul", "div", "sum")]
saved += [(torch, n, _wrap(torch, n)) for n in ("empty", "zeros", "empty_like", "zeros_like", "ones_like", "tensor")]
step()
torch.cuda.synchronize()
for owner, n, orig in saved:
    setattr(owner, n, orig)
print("python-level call sites in ONE step (method, file:line, count):")
for (n, site), c in sites.most_common(60):
    print("%4d  %-12s %s" % (c, n, site))
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ops, where = Counter(), {}
for ev in prof.events():
    n = ev.name
    if not (n.startswith("aten::") or "emcpy" in n or "emset" in n):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue                                      # nested operator: counted with its parent
    frames = [f for f in (ev.stack or []) if "mindtheedge_amd" in f or "bench.py" in f or "tools/" in f]
    site = frames[0].split("/root/repo/")[-1] if frames else "?"
    ops[(n, site)] += 1
print("top-level ATen operators / copies in ONE training step (B = %d): %d" % (B, sum(ops.values())))
for (n, site), c in ops.most_common(70):
    print("%4d  %-28s %s" % (c, n, site))
kern = Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        kern[ev.name[:60]] += 1
print("device activities:", sum(kern.values()))
for n, c in kern.most_common(25):
    print("%4d  %s" % (c, n))

# ---- host profile of three steps (cProfile, cumulative time per function of this package)
import cProfile  # noqa: E402
import pstats  # noqa: E402
import io  # noqa: E402
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
sio = io.StringIO()
pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(45)
print("\n".join(l for l in sio.getvalue().splitlines() if l.strip())[:9000])
