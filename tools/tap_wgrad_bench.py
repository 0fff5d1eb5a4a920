#!/usr/bin/env python3
"""development aid (round 5): the one-channel 3x3 weight gradients (InvDepth head, rank-1 column of the iconv layers) at their T8 shapes:
matrix-core GEMM over pixels (tap_wgrad.hip) against the VALU kernels (mte_debug_set(31, 0)).  us per call, GB/s of the activation read."""
import os, sys
os.environ.setdefault("MTE_USE_DEV_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindtheedge_amd import kernels as K
K.set_compute_dtype("bf16")
B = 8


def best(f):
    for _ in range(3): f()
    t = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        t = min(t, e0.elapsed_time(e1) * 100)
    return t


for C, H, W in ((32, 384, 1280), (64, 192, 640), (128, 96, 320), (256, 48, 160)):
    x = K.new_act(B, C, H, W).normal_()
    dl = torch.randn(B, H, W, device="cuda")
    dwb = torch.empty(C * 9 + 1, device="cuda")
    rec = torch.empty((int(K.lib.mte_invdepth_bwd_weight_workspace_elems(C)),), dtype=torch.float32, device="cuda")
    xp, ldx = K._pl(x)
    mb = B * H * W * C * 2 / 1e6
    line = "head  C %3d @%dx%d (%.0f MB)" % (C, H, W, mb)
    for name, knob in (("valu", 0), ("mfma", 1)):
        K.lib.mte_debug_set(31, knob)
        t = best(lambda: K.lib.mte_invdepth_bwd_weight(xp, ldx, dl.data_ptr(), dwb.data_ptr(), rec.data_ptr(), B, H, W, C, K.DT_BF16, K._stream()))
        line += "  %s %6.1f us %5.0f GB/s" % (name, t, mb / t * 1e3)
    print(line)
for N, H, W in ((32, 384, 1280), (64, 192, 640), (128, 96, 320)):
    dy = K.new_act(B, N, H, W).normal_()
    inv = torch.rand(B, H // 2, W // 2, device="cuda")
    dw = torch.empty(N, 9, device="cuda")
    rec = torch.empty((int(K.lib.mte_rank1_conv_bwd_records_elems(N)),), dtype=torch.float32, device="cuda")
    dp, ld = K._pl(dy)
    mb = B * H * W * N * 2 / 1e6
    line = "rank1 N %3d @%dx%d (%.0f MB)" % (N, H, W, mb)
    for name, knob in (("valu", 0), ("mfma", 1)):
        K.lib.mte_debug_set(31, knob)
        t = best(lambda: K.lib.mte_rank1_conv_bwd_weight(dp, ld, inv.data_ptr(), dw.data_ptr(), 9, rec.data_ptr(), B, H // 2, W // 2, N, K.DT_BF16, K._stream()))
        line += "  %s %6.1f us %5.0f GB/s" % (name, t, mb / t * 1e3)
    print(line)
K.lib.mte_debug_set(31, 1)
