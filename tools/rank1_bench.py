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
