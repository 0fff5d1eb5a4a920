#!/usr/bin/env python3
"""development aid (round 6): race screen for the BACKWARD pass at the benchmark geometry -- N times forward + backward of the same batch from the same weights
(dropout off, no flip); per repetition the parameter gradients are compared with the first one.  The backward is NOT bit-reproducible as a whole: the GroupNorm
backward's group sums are float-atomic, so every data gradient upstream of a norm differs in its last bits from run to run (each weight-gradient KERNEL is a
fixed-order sum: tests/test_gpu_determinism.py).  What a race looks like is an O(1) difference in some tensor: the screen reports the largest relative differences
(to the gradient's own maximum); measured on the final round-6 tree: conv weights <= 9e-5, vector parameters (sums with cancellation) <= 2e-2.  usage: train_repro.py [N]"""
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
K.set_compute_dtype("bf16")
torch.manual_seed(42)
net = PackNetSAN01(dropout=None, version="1A").to(dev)
model = SemiSupEdgeModel(supervised_loss_weight=1.0, depth_edges_loss_weight=1.0, supervised_method="sparse-silog",
                         supervised_num_scales=1, edges_depth_edge_loss_all_scales=True, flip_lr_prob=0.0)
model.add_depth_net(net)
model.add_edge_loss(GradLoss("cross_entropy", True, [], 10.0, 1.0))
batch = bench.device_batch(8, 384, 1280, seed=1234, device=dev)
random.seed(100)
model.train()
flat = FlatParameters(net.parameters())
opt = FusedAdam(flat, lr=0.0, reducer=None)
name_of = {id(p): n for n, p in net.named_parameters()}
ref, worst_exact, soft, bad = None, [], [], 0
for rep in range(N):
    opt.zero_grad()
    out = model(batch)
    out["loss"].backward()
    K.join_side_stream()
    torch.cuda.synchronize()
    K.check_device_errors()
    g = flat.grad.clone()
    if ref is None:
        ref, loss0 = g, float(out["loss"])
        continue
    for p, o in zip(flat.params, flat.offsets):
        n = name_of.get(id(p), "?")
        a, b = g[o:o + p.numel()], ref[o:o + p.numel()]
        if torch.equal(a, b):
            continue
        rel = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        if p.dim() >= 4:
            bad += 1
            worst_exact.append((rep, n, rel))
        else:
            soft.append((rel, n, float(b.abs().max())))
soft.sort(reverse=True)
worst_exact.sort(key=lambda t: -t[2])
print("%d repetitions at B = 8, 384x1280: loss %.6f; conv / conv3d weight gradients not bit-equal to the first repetition: %d, largest relative differences %s"
      % (N, loss0, bad, [(n, "%.1e" % r) for _, n, r in worst_exact[:5]]))
print("RACE SUSPECT" if worst_exact and worst_exact[0][2] > 1e-3 else "no conv weight gradient differs by more than 1e-3 of its maximum")
print("vector parameters (biases, GroupNorm gamma / beta: float atomics) -- largest relative differences (rel to the gradient's max, name, max |g|):")
for rel, n, m in soft[:8]:
    print("   %.2e  %-48s %.3e" % (rel, n, m))
