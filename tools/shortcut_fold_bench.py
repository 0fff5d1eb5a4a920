#!/usr/bin/env python3
"""development aid (round 4): data gradient of a residual block's input at the T8 shapes -- 1x1 launch + accumulating 3x3 launch against the one-launch
form (mte_conv2d_patch_fwd_plus1x1), and the plain 3x3 alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindtheedge_amd import kernels as K
K.set_compute_dtype("bf16")
lib = K.lib
B = 8
for c1, cp, c2, H, W in ((64, 32, 64, 192, 640), (64, 64, 64, 192, 640)):
    w1 = torch.randn(c1, cp, 3, 3, device="cuda") * 0.05
    w3 = torch.randn(c2, cp, 1, 1, device="cuda") * 0.05
    dy1 = K.new_act(B, c1, H, W).normal_(); dy3 = K.new_act(B, c2, H, W).normal_()
    p1, p3 = K.WeightPack(), K.WeightPack()
    b1, b3 = p1.get_patch(w1, 'b').data_ptr(), p3.get_patch(w3, 'b').data_ptr()
    dx = K.new_act(B, cp, H, W)
    a1, l1 = K._pl(dy1); a3, l3 = K._pl(dy3); dp, dl = K._pl(dx)
    st = K._stream

    def two():
        lib.mte_conv2d_patch_fwd(a3, l3, b3, 0, dp, dl, B, H, W, c2, cp, 1, 1, 0, st())
        lib.mte_conv2d_patch_fwd(a1, l1, b1, 0, dp, dl, B, H, W, c1, cp, 3, 3, 1, st())
    fns = {"1x1 + accumulating 3x3": two,
           "one launch": lambda: lib.mte_conv2d_patch_fwd_plus1x1(a1, l1, b1, 0, dp, dl, B, H, W, c1, cp, a3, l3, b3, c2, st()),
           "3x3 alone": lambda: lib.mte_conv2d_patch_fwd(a1, l1, b1, 0, dp, dl, B, H, W, c1, cp, 3, 3, 0, st()),
           "1x1 alone": lambda: lib.mte_conv2d_patch_fwd(a3, l3, b3, 0, dp, dl, B, H, W, c2, cp, 1, 1, 0, st())}
    line = "dy1 %d + dy3 %d -> dx %d @%dx%d" % (c1, c2, cp, H, W)
    for name, f in fns.items():
        for _ in range(3): f()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100)
        line += "  %s %6.1f us" % (name, best)
    print(line)
