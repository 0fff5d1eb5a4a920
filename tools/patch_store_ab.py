#!/usr/bin/env python3
"""development aid (round 4): same-process A/B of mte_conv2d_patch_fwd between two builds of the library (interleaved launches, best of many):
usage: patch_store_ab.py OTHER_LIB.so     -- the store loop of the LDS-patch forward, plain and accumulating"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindtheedge_amd import kernels as K
K.set_compute_dtype("bf16")
other = ctypes.CDLL(sys.argv[1])
fo = other.mte_conv2d_patch_fwd
fn = K.lib.load().mte_conv2d_patch_fwd
for f in (fo,):
    f.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long] + [ctypes.c_int] * 8 + [ctypes.c_void_p]
    f.restype = ctypes.c_int
B = 8
for cm, N, H, W, k in ((64, 32, 384, 1280, 3), (96, 64, 192, 640, 3), (32, 32, 384, 1280, 7), (64, 64, 192, 640, 3), (32, 64, 384, 1280, 3), (64, 64, 192, 640, 1), (32, 64, 192, 640, 1), (128, 64, 96, 320, 3), (64, 32, 192, 640, 5)):
    x = K.new_act(B, cm, H, W).normal_()
    wt = torch.randn(N, cm, k, k, device="cuda") * 0.05
    bias = torch.zeros(N, device="cuda")
    pack = K.WeightPack(); pack.get(wt, x.dtype, False)
    pf = pack.get_patch(wt, 'f').data_ptr()
    y = K.new_act(B, N, H, W).zero_()
    xp, ldx = K._pl(x); yp, ldy = K._pl(y)
    line = "%3d -> %2d k%d @%dx%d" % (cm, N, k, H, W)
    for acc in (0, 1):
        best = {"other": 1e9, "this": 1e9}
        for rep in range(12):
            for name, f in (("other", fo), ("this", fn)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f(xp, ldx, pf, bias.data_ptr(), yp, ldy, B, H, W, cm, N, k, k, acc, K._stream())
                e0.record()
                for _ in range(5): f(xp, ldx, pf, bias.data_ptr(), yp, ldy, B, H, W, cm, N, k, k, acc, K._stream())
                e1.record(); torch.cuda.synchronize()
                best[name] = min(best[name], e0.elapsed_time(e1) * 200)
        line += "  %s other %6.1f this %6.1f us" % ("accumulate" if acc else "plain", best["other"], best["this"])
    print(line)
