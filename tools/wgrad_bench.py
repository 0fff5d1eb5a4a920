#!/usr/bin/env python3
"""development aid: weight-gradient kernels at the training shapes with 65..128 output channels: the wide LDS-patch variant against the
generic per-tap kernel (patch kernels switched off), time per call incl. the unpack pass (HIP events over 20 calls) and agreement."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402

B = 8
for cin, cout, k, H, W in ((128, 128, 3, 96, 320), (200, 128, 3, 96, 320), (64, 128, 3, 96, 320), (256, 128, 3, 48, 160), (384, 256, 3, 48, 160)):
    g = torch.Generator().manual_seed(1)
    x = K.new_act(B, K.round8(cin), H, W)
    x.copy_(torch.randn(B, K.round8(cin), H, W, generator=g).cuda())
    dy = K.new_act(B, cout, H, W)
    dy.copy_(torch.randn(B, cout, H, W, generator=g).cuda())
    w = torch.empty(cout, cin, k, k, device="cuda")
    out = {}
    for patch in (True, False):
        K.use_patch_kernels(patch)
        for _ in range(3):
            dw, _ = K._conv_wgrad(x, dy, w, False, None, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dw, _ = K._conv_wgrad(x, dy, w, False, None, None)
        e1.record()
        torch.cuda.synchronize()
        out[patch] = (e0.elapsed_time(e1) / 20, dw.clone())
    K.use_patch_kernels(True)
    fl = 2.0 * B * H * W * cin * cout * k * k
    print("%4d -> %-4d k%d @%dx%-4d  patch %7.1f us %6.1f TF   generic %7.1f us %6.1f TF   max rel diff %.1e" % (
        cin, cout, k, H, W, out[True][0] * 1e3, fl / out[True][0] / 1e9, out[False][0] * 1e3, fl / out[False][0] / 1e9,
        float((out[True][1] - out[False][1]).abs().max() / out[False][1].abs().max())))
