#!/usr/bin/env python3
"""development aid (round 4): the three rank-1 kernels of the decoder iconv layers (mte_rank1_conv_fwd / _bwd_data / _bwd_weight) at their T8 shapes: us and TB/s of the big tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindtheedge_amd import kernels as K
K.set_compute_dtype("bf16")
lib = K.lib
B = 8
for N, h, w in ((32, 192, 640), (64, 96, 320), (128, 48, 160)):
    dy = K.new_act(B, N, 2 * h, 2 * w).normal_()
    inv = torch.rand(B, 1, h, w, device="cuda")
    wt = torch.randn(N, 65, 3, 3, device="cuda")
    dinv = torch.empty(B, 1, h, w, device="cuda")
    y = K.new_act(B, N, 2 * h, 2 * w)
    gw = torch.empty_like(wt)
    rec = torch.empty(int(lib.mte_rank1_conv_bwd_records_elems(N)), device="cuda")
    dp, ld = K._pl(dy); yp, ldy = K._pl(y)
    fns = {"fwd": lambda: lib.mte_rank1_conv_fwd(inv.data_ptr(), wt.data_ptr() + 4 * 64 * 9, 65 * 9, yp, ldy, B, h, w, N, K._dt(y), K._stream()),
           "bwd_data": lambda: lib.mte_rank1_conv_bwd_data(dp, ld, wt.data_ptr() + 4 * 64 * 9, 65 * 9, dinv.data_ptr(), B, h, w, N, 0, K._dt(dy), K._stream()),
           "bwd_weight": lambda: lib.mte_rank1_conv_bwd_weight(dp, ld, inv.data_ptr(), gw.data_ptr() + 4 * 64 * 9, 65 * 9, rec.data_ptr(), B, h, w, N, K._dt(dy), K._stream())}
    mb = B * 4 * h * w * N * 2 / 1e6
    line = "N %3d @%dx%d (%.0f MB)" % (N, 2 * h, 2 * w, mb)
    for name, f in fns.items():
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        line += "  %s %6.1f us %4.2f TB/s" % (name, us, mb / us)
    print(line)

# the forward term inside the LDS-patch conv (mte_conv2d_patch_fwd_rank1) against the two-launch form (mte_rank1_conv_fwd, then the accumulating conv), and the plain conv
print("forward of the layer: map term written first + accumulating conv | term in the conv's store loop | (the conv alone)")
for cm, N, H, W in ((64, 32, 384, 1280), (96, 64, 192, 640)):
    x = K.new_act(B, cm, H, W).normal_()
    inv = torch.rand(B, 1, H // 2, W // 2, device="cuda")
    wt = torch.randn(N, cm + 1, 3, 3, device="cuda") * 0.05
    bias = torch.zeros(N, device="cuda")
    wm = wt[:, :cm].contiguous()
    pack = K.WeightPack(); pack.get(wm, x.dtype, False)
    pf = pack.get_patch(wm, 'f').data_ptr()
    y = K.new_act(B, N, H, W)
    xp, ldx = K._pl(x); yp, ldy = K._pl(y)
    w1 = wt.data_ptr() + 4 * cm * 9

    def two():
        lib.mte_rank1_conv_fwd(inv.data_ptr(), w1, (cm + 1) * 9, yp, ldy, B, H // 2, W // 2, N, K._dt(y), K._stream())
        lib.mte_conv2d_patch_fwd(xp, ldx, pf, bias.data_ptr(), yp, ldy, B, H, W, cm, N, 3, 3, 1, K._stream())
    fns = {"two launches": two,
           "fused": lambda: lib.mte_conv2d_patch_fwd_rank1(xp, ldx, pf, bias.data_ptr(), yp, ldy, B, H, W, cm, N, inv.data_ptr(), w1, (cm + 1) * 9, K._stream()),
           "conv alone": lambda: lib.mte_conv2d_patch_fwd(xp, ldx, pf, bias.data_ptr(), yp, ldy, B, H, W, cm, N, 3, 3, 0, K._stream())}
    line = "%3d -> %2d @%dx%d" % (cm, N, H, W)
    for name, f in fns.items():
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        line += "  %s %6.1f us" % (name, e0.elapsed_time(e1) * 100)
    print(line)
