#!/usr/bin/env python3
"""Round-4 study (verdict item 4): does the bf16 benchmark mode TRAIN like the fp32 validation mode?

  (i)   gradient of the full loss (silog + depth-edge loss, all four scales) at the random-init operating point, 384x1280:
        cosine and norm ratio of the bf16-mode gradient against the fp32-mode gradient (fp32 mode is pinned to the CPU oracle
        at 1e-3 per element by tests/test_gpu_oracle_fullsize.py), whole gradient and per tensor;
  (ii)  `steps` optimizer steps (FusedAdam, lr 1e-4) over one fixed set of synthetic batches in both modes -- same seeds, no
        dropout, no flip: loss curves, and the cosine / norm ratio of the parameter displacement (theta_t - theta_0) every 50 steps;
  (iii) the gradient comparison of (i) again at the trained point (both modes evaluated at the fp32 run's final parameters).

usage: bf16_fidelity.py [steps [batch [H W]]]  ->  prints the report (tools/... > profiles/r04_bf16_training_fidelity.txt)"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (384, 1280)
NB = 4                                   # batches in the fixed set, visited round-robin
EVERY = 50

from mindtheedge_amd import kernels as K  # noqa: E402
from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01  # noqa: E402
from mindtheedge_amd.models.SemiSupEdgeModel import SemiSupEdgeModel  # noqa: E402
from mindtheedge_amd.losses.grad_loss import GradLoss  # noqa: E402
from mindtheedge_amd.trainers.data_parallel import FlatParameters, FusedAdam  # noqa: E402

dev = torch.device("cuda", 0)


def batch_of(seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, device=dev)
    # a smooth scene instead of white noise: depth = a few random planes / blobs, so that the edge loss has structure to learn
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device=dev), torch.linspace(0, 1, W, device=dev), indexing="ij")
    depth = torch.zeros(B, 1, H, W, device=dev)
    rgb = torch.zeros(B, 3, H, W, device=dev)
    for b in range(B):
        d = 5.0 + 40.0 * yy.flip(0) * float(r(1)) + 10.0 * xx * float(r(1))
        for _ in range(6):
            cx, cy, rad, dd = float(r(1)), float(r(1)), 0.05 + 0.15 * float(r(1)), 3.0 + 30.0 * float(r(1))
            m = ((xx - cx) ** 2 + ((yy - cy) * H / W) ** 2) < rad ** 2
            d = torch.where(m, torch.full_like(d, dd), d)
        depth[b, 0] = d
        base = r(3, 1, 1)
        rgb[b] = (base * (1.0 / (1.0 + 0.05 * d))[None] + 0.05 * r(3, H, W)).clamp(0, 1)
    batch = {"rgb": rgb, "depth": depth * (r(B, 1, H, W) < 0.05).float()}
    for s in range(4):
        sfx = "" if s == 0 else "_%d" % s
        ds = depth[:, :, ::1 << s, ::1 << s]
        gx = (ds[:, :, :, 1:] - ds[:, :, :, :-1]).abs()
        gy = (ds[:, :, 1:, :] - ds[:, :, :-1, :]).abs()
        e = torch.zeros_like(ds)
        e[:, :, :, 1:] += (gx > 1.0).float()
        e[:, :, 1:, :] += (gy > 1.0).float()
        batch["edge" + sfx] = e.clamp(max=1.0)
        batch["normal" + sfx] = (r(*ds.shape) * 2 - 1) * math.pi
    return batch


def build(mode, flat_init=None):
    K.set_compute_dtype(mode)
    K.set_grad_sink(None)
    torch.manual_seed(42)
    net = PackNetSAN01(dropout=None, version="1A").to(dev)
    model = SemiSupEdgeModel(supervised_loss_weight=1.0, depth_edges_loss_weight=1.0, supervised_method="sparse-silog",
                             supervised_num_scales=1, edges_depth_edge_loss_all_scales=True, flip_lr_prob=0.0)
    model.add_depth_net(net)
    model.add_edge_loss(GradLoss("cross_entropy", True, [], 10.0, 1.0))
    model.train()
    flat = FlatParameters(net.parameters())
    if flat_init is not None:
        flat.flat.copy_(flat_init)
        K.bump_weights_epoch()
    return net, model, flat


def gradient(mode, batch, flat_init=None):
    net, model, flat = build(mode, flat_init)
    flat.zero_grad()
    out = model(batch)
    out["loss"].sum().backward()
    K.join_side_stream()
    torch.cuda.synchronize()
    g = flat.grad.clone()
    names = [(n, flat.offset_of[id(p)], p.numel()) for n, p in net.named_parameters() if id(p) in flat.offset_of]
    loss = float(out["loss"].detach().sum())
    K.set_grad_sink(None)
    return g, names, loss


def cos_ratio(a, b):
    a, b = a.double(), b.double()
    return float((a * b).sum() / (a.norm() * b.norm()).clamp(min=1e-300)), float(a.norm() / b.norm().clamp(min=1e-300))


def compare_gradients(tag, batch, flat_init=None):
    g32, names, l32 = gradient("fp32", batch, flat_init)
    g16, _, l16 = gradient("bf16", batch, flat_init)
    c, r = cos_ratio(g16, g32)
    per = []
    for n, o, k in names:
        if float(g32[o:o + k].abs().max()) > 0:
            per.append((cos_ratio(g16[o:o + k], g32[o:o + k]), n, k))
    per.sort()
    cs = sorted(p[0][0] for p in per)
    print("%s: loss fp32 %.6f  bf16 %.6f (rel %.2e) | whole gradient (%d tensors, %.1f M elements): cosine %.4f, norm ratio bf16/fp32 %.4f"
          % (tag, l32, l16, abs(l16 - l32) / abs(l32), len(per), g32.numel() / 1e6, c, r))
    print("    per tensor cosine: min %.4f (%s)  5%% %.4f  median %.4f  95%% %.4f | tensors with cosine < 0.9: %d"
          % (cs[0], per[0][1], cs[len(cs) // 20], cs[len(cs) // 2], cs[-1 - len(cs) // 20], sum(1 for v in cs if v < 0.9)))
    return c, r


def train(mode, batches):
    net, model, flat = build(mode)
    opt = FusedAdam(flat, lr=1e-4)
    theta0 = flat.flat.clone()
    losses, snaps = [], {}
    for t in range(steps):
        opt.zero_grad()
        out = model(batches[t % NB])
        out["loss"].backward()
        opt.step()
        losses.append(out["loss"].detach().sum())
        if (t + 1) % EVERY == 0 or t + 1 == steps:
            snaps[t + 1] = (flat.flat - theta0).clone()
    torch.cuda.synchronize()
    final = flat.flat.clone()
    K.set_grad_sink(None)
    return [float(v) for v in losses], snaps, final


print("bf16 training fidelity: PackNetSAN01 + SemiSupEdgeModel, B = %d, %dx%d, %d batches round-robin, %d Adam steps (lr 1e-4), no dropout, no flip" % (B, H, W, NB, steps))
batches = [batch_of(1234 + i) for i in range(NB)]
print("\n(i) gradient at the random-init point (xavier, seed 42), batch 0")
compare_gradients("    init", batches[0])
print("\n(ii) training runs")
l32, s32, final32 = train("fp32", batches)
l16, s16, final16 = train("bf16", batches)
# control: the SAME mode twice.  The backward pass is not bit-reproducible (fp32 atomics in a few weight-gradient tails), and Adam divides
# by sqrt(v): where a gradient element is noise-sized its +-lr step follows the noise -- how far do two identical runs drift apart?
l32b, s32b, _ = train("fp32", batches)
print("    step   loss fp32    loss bf16    rel diff | displacement theta_t - theta_0: cosine(bf16, fp32)  norm ratio bf16/fp32  |fp32 displacement|")
for t in sorted(s32):
    c, r = cos_ratio(s16[t], s32[t])
    a32 = sum(l32[max(0, t - NB):t]) / min(NB, t)
    a16 = sum(l16[max(0, t - NB):t]) / min(NB, t)
    print("    %4d   %.6f   %.6f   %+.2e |                                   %.4f               %.4f                 %.4e"
          % (t, a32, a16, (a16 - a32) / abs(a32), c, r, float(s32[t].double().norm())))
print("    control, fp32 run against a SECOND fp32 run (same seeds; the backward's atomics order is the only difference):")
for t in sorted(s32):
    c, r = cos_ratio(s32b[t], s32[t])
    a32 = sum(l32[max(0, t - NB):t]) / min(NB, t)
    a32b = sum(l32b[max(0, t - NB):t]) / min(NB, t)
    print("    %4d   %.6f   %.6f   %+.2e |                                   %.4f               %.4f" % (t, a32, a32b, (a32b - a32) / abs(a32), c, r))
print("    (losses: mean over the last %d steps = one pass over the batch set; first pass fp32 %.6f / bf16 %.6f)" % (NB, sum(l32[:NB]) / NB, sum(l16[:NB]) / NB))
worst = max(abs(a - b) / abs(a) for a, b in zip(l32, l16))
print("    largest per-step loss difference over the run: %.2e relative" % worst)
print("\n(iii) gradient at the trained point (fp32 run's parameters after %d steps), batch 0" % steps)
compare_gradients("    trained", batches[0], final32)
K.set_compute_dtype("bf16")
