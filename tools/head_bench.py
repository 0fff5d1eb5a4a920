#!/usr/bin/env python3
"""development aid (round 4): the InvDepth head forward at its four T8 shapes, matrix-core form against the VALU row-marching kernel (mte_debug_set(30, 0))."""
import os, sys
os.environ.setdefault("MTE_USE_DEV_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindtheedge_amd import kernels as K
K.set_compute_dtype("bf16")
B = 8
for C, H, W in ((32, 384, 1280), (64, 192, 640), (128, 96, 320), (256, 48, 160)):
    x = K.new_act(B, C, H, W).normal_()
    w = torch.randn(1, C, 3, 3, device="cuda") * 0.05
    b = torch.zeros(1, device="cuda")
    out = torch.empty(B, H, W, device="cuda")
    xp, ldx = K._pl(x)
    line = "C %3d @%dx%d (%.0f MB)" % (C, H, W, B * H * W * C * 2 / 1e6)
    for name, knob in (("valu", 0), ("product rule", 1), ("mfma", 2), ("pf", 6), ("pf r16", 6 + (16 << 8)), ("pf r32", 6 + (32 << 8)), ("r16", 2 + (16 << 8))):
        K.lib.mte_debug_set(30, knob)
        f = lambda: K.lib.mte_invdepth_fwd(xp, ldx, w.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, C, 0.5, K.DT_BF16, K._stream())
        for _ in range(3): f()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        line += "  %s %5.1f" % (name, best)
    print(line)
K.lib.mte_debug_set(30, 1)
