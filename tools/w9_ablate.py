#!/usr/bin/env python3
"""development aid (round 5): what a K-step of the nine-tap weight gradient (csrc/conv_wgrad9.hip) costs -- private builds with pieces of the
main loop left out (-DMTE_W9_ABL=bits: 1 no LDS-DMA, 2 no MFMAs, 4 no fragment reads, 8 no barrier, 16 no epilogue stores), kernel-only time by HIP
events around mte_conv2d_wgrad alone (no unpack pass).  usage: w9_ablate.py [bits ...]   (B = 8)"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mindtheedge_amd import _build  # noqa: E402

_build.build()
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402

variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 4, 8, 16, 3, 6, 7]
shapes = [(512, 512, 24, 80), (4096, 256, 24, 80), (256, 256, 48, 160)]
libs = {}
for v in variants:
    so, obj = "/tmp/libmte_w9abl%d.so" % v, "/tmp/conv_wgrad9_abl%d.o" % v
    subprocess.check_call([_build._hipcc()] + _build.FLAGS + ["-DMTE_W9_ABL=%d" % v, "-I", _build.CSRC, "-c", os.path.join(_build.CSRC, "conv_wgrad9.hip"), "-o", obj])
    others = [o for o in glob.glob(os.path.join(_build.CSRC, "*.o")) if not o.endswith("conv_wgrad9.o")]
    subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-o", so, obj] + others)
    libs[v] = ctypes.CDLL(so)
B = 8
for cin, cout, H, W in shapes:
    g = torch.Generator().manual_seed(1)
    x = K.new_act(B, cin, H, W); x.copy_(torch.randn(B, cin, H, W, generator=g).cuda())
    dy = K.new_act(B, cout, H, W); dy.copy_(torch.randn(B, cout, H, W, generator=g).cuda())
    cap = max(1, min(64, (96 << 20) // (4 * cout * 9 * cin)))
    stage = torch.empty((cap + 32, cout, 9, cin), dtype=torch.float32, device="cuda")
    parts = ctypes.c_int(1)
    fl = 2.0 * B * H * W * cin * cout * 9
    row = []
    for v in variants:
        f = libs[v].mte_conv2d_wgrad
        f.restype = ctypes.c_int
        args = (ctypes.c_void_p(x.data_ptr()), ctypes.c_long(cin), ctypes.c_void_p(dy.data_ptr()), ctypes.c_long(cout), ctypes.c_void_p(stage.data_ptr()), ctypes.c_int(cap),
                ctypes.byref(parts), B, H, W, cin, cout, 3, 3, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        for _ in range(3):
            assert f(*args) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f(*args)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row.append("abl %2d: %6.1f us" % (v, us))
    print("%4d -> %-4d @%dx%-3d parts %2d  %5.0f GFLOP | %s" % (cin, cout, H, W, parts.value, fl / 1e9, " | ".join(row)))
