#!/usr/bin/env python3
"""development aid (round 4): what does `accumulate` cost the implicit-GEMM epilogue?  plain vs accumulating launch of mte_conv2d_igemm at the step's shapes
(optionally against another build: igemm_accum_ab.py OTHER_LIB.so)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindtheedge_amd import kernels as K
K.set_compute_dtype("bf16")
libs = [("this", K.lib.load().mte_conv2d_igemm)]
if len(sys.argv) > 1:
    o = ctypes.CDLL(sys.argv[1]).mte_conv2d_igemm
    o.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    o.restype = ctypes.c_int
    libs.insert(0, ("other", o))
B = 8
for cm, N, H, W, k in ((256, 256, 48, 160, 3), (512, 512, 24, 80, 3), (128, 128, 96, 320, 3), (256, 128, 48, 160, 3), (192, 128, 96, 320, 3), (512, 256, 24, 80, 3), (64, 96, 192, 640, 3), (128, 128, 96, 320, 1)):
    x = K.new_act(B, cm, H, W).normal_()
    wt = torch.randn(N, cm, k, k, device="cuda") * 0.05
    bias = torch.zeros(N, device="cuda")
    pack = K.WeightPack(); wf, _ = pack.get(wt, x.dtype, False)
    y = K.new_act(B, N, H, W).zero_()
    xp, ldx = K._pl(x); yp, ldy = K._pl(y)
    ws, ws_n = K._splitk_workspace(B * H * W, N, x.device)
    line = "%3d -> %3d k%d @%dx%d" % (cm, N, k, H, W)
    for acc in (0, 1):
        best = {n: 1e9 for n, _ in libs}
        for rep in range(10):
            for name, f in libs:
                call = lambda: f(xp, ldx, wf.data_ptr(), bias.data_ptr(), yp, ldy, 0, B, H, W, cm, N, k, k, K._dt(x), K._ptr(ws), ws_n, acc | 2, K._stream())
                call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): call()
                e1.record(); torch.cuda.synchronize()
                best[name] = min(best[name], e0.elapsed_time(e1) * 200)
        line += "  %s " % ("accumulate" if acc else "plain") + " ".join("%s %6.1f" % (n, best[n]) for n, _ in libs) + " us"
    print(line)
