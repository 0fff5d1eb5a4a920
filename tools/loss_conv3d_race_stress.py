#!/usr/bin/env python3
"""development aid (round 4): race screen for the kernels that hand data between workgroups or LDS buffers without a kernel boundary --
the fused loss forward (records -> ticket -> image sums -> ticket -> losses) and the persistent, double-buffered conv3d forward: repeated
launches, optionally with a copy + GEMM on another queue, must reproduce the first result bit for bit.  usage: loss_conv3d_race_stress.py [reps [noise]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
noise = torch.cuda.Stream() if len(sys.argv) > 2 else None
na = torch.empty(256 << 20, dtype=torch.uint8, device="cuda"); nb = torch.empty_like(na)
nm = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16); nmo = torch.empty_like(nm)


def disturb():
    if noise is not None:
        with torch.cuda.stream(noise):
            nb.copy_(na)
            torch.mm(nm, nm, out=nmo)


bad = 0
# ---- fused loss: forward values + gradients of all four scales
import test_gpu_edge_loss_fused as T  # noqa: E402
for B, H, W in ((8, 384, 1280), (3, 96, 160), (2, 200, 328)):
    invs, batch = T._maps(B, H, W, seed=B)
    m = T._model(True)
    dev = torch.device("cuda")
    b = {k: v.to(dev) for k, v in batch.items()}
    first = None
    miss = 0
    for r in range(reps):
        xs = [i.to(dev).requires_grad_(True) for i in invs]
        disturb()
        edge, sup = m._fused_losses(xs, b)
        (sup + edge).sum().backward()
        res = [edge.detach().clone(), sup.detach().clone()] + [x.grad for x in xs]
        torch.cuda.synchronize()
        if first is None:
            first = res
        elif not all(torch.equal(a, c) for a, c in zip(first, res)):
            miss += 1
    bad += miss
    print("fused loss B%d %dx%-4d  %d / %d repetitions differ" % (B, H, W, miss, reps))
# ---- conv3d unpack forward / data gradient on the matrix cores (persistent tile walk, two LDS buffers)
g = torch.Generator().manual_seed(2)
for C, B, H, W in ((32, 8, 192, 640), (64, 8, 96, 320), (32, 2, 50, 70)):
    x = K.image_to_act((torch.rand(B, C, H, W, generator=g) * 2 - 1).cuda())
    dout = K.image_to_act((torch.rand(B, C, 2 * H, 2 * W, generator=g) * 2 - 1).cuda())
    w3 = ((torch.rand(4, 1, 3, 3, 3, generator=g) - 0.5) * 0.8).cuda()
    b3 = ((torch.rand(4, generator=g) - 0.5) * 0.4).cuda()
    out, dx = K.new_act(B, C, 2 * H, 2 * W), K.new_act(B, C, H, W)
    first = None
    miss = 0
    for r in range(reps):
        disturb()
        K.lib.mte_unpack3d_fwd(*K._pl(x), w3.data_ptr(), b3.data_ptr(), *K._pl(out), B, H, W, C, K._dt(x), K._stream())
        K.lib.mte_unpack3d_bwd_data(*K._pl(dout), w3.data_ptr(), *K._pl(dx), B, H, W, C, K._dt(x), K._stream())
        torch.cuda.synchronize()
        res = [out.clone(), dx.clone()]
        if first is None:
            first = res
        elif not all(torch.equal(a, c) for a, c in zip(first, res)):
            miss += 1
        out.fill_(3.0); dx.fill_(3.0)
    bad += miss
    print("conv3d unpack C%d B%d %dx%-4d  %d / %d repetitions differ" % (C, B, H, W, miss, reps))
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
