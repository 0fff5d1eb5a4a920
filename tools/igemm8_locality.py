#!/usr/bin/env python3
"""development aid (round 5): what would L2 locality of the nine taps be worth to the 8-phase implicit GEMM?  A private build of conv_igemm8.hip in which
every tap reads the CENTRE pixel (-DMTE_I8_TAP0: same instruction stream, same bytes through LDS-DMA, but the nine tap sweeps of an input slice hit the
lines the first one fetched) against the shipped kernel, forward launches at the T8 shapes, interleaved.  usage: igemm8_locality.py"""
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

libs = {}
for tag, defs in (("shipped", []), ("tap0", ["-DMTE_I8_TAP0"])):
    so, obj = "/tmp/libmte_i8%s.so" % tag, "/tmp/conv_igemm8_%s.o" % tag
    subprocess.check_call([_build._hipcc()] + _build.FLAGS + defs + ["-I", _build.CSRC, "-c", os.path.join(_build.CSRC, "conv_igemm8.hip"), "-o", obj])
    others = [o for o in glob.glob(os.path.join(_build.CSRC, "*.o")) if not o.endswith("conv_igemm8.o")]
    subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-o", so, obj] + others)
    libs[tag] = ctypes.CDLL(so)
B = 8
for cin, cout, H, W in ((256, 256, 48, 160), (512, 512, 24, 80), (384, 256, 48, 160), (128, 256, 48, 160)):
    g = torch.Generator().manual_seed(1)
    x = K.new_act(B, cin, H, W); x.copy_(torch.randn(B, cin, H, W, generator=g).cuda())
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
    wf, _ = K.WeightPack().get(w, x.dtype, False)
    y = K.new_act(B, cout, H, W)
    b = torch.zeros(cout, device="cuda")
    res = {}
    for rnd in range(3):
        for tag, lib in libs.items():
            f = lib.mte_conv2d_igemm
            f.restype = ctypes.c_int
            args = (ctypes.c_void_p(x.data_ptr()), ctypes.c_long(cin), ctypes.c_void_p(wf.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_long(cout),
                    0, B, H, W, cin, cout, 3, 3, 0, ctypes.c_void_p(0), ctypes.c_long(0), 2, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            for _ in range(3):
                assert f(*args) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f(*args)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(tag, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * B * H * W * cin * cout * 9
    print("%4d -> %-4d @%dx%-3d | %s" % (cin, cout, H, W, " | ".join("%s %6.1f us (%5.0f TF)" % (t, min(v), fl / min(v) / 1e6) for t, v in res.items())))
