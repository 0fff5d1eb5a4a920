#!/usr/bin/env python3
"""development aid (round 4): where a wave of the 8-phase implicit GEMM (csrc/conv_igemm8.hip, tile-per-workgroup form) spends a K-tile:
s_memtime sums per phase -- load section (fragment reads + LDS-DMA issue + address work) | first barrier | MFMAs | second barrier.
Private diagnostic build (-DMTE_STAMPS) under /tmp.   usage: igemm8_stamps.py cin,cout,k,H,W ...  (B = 8, forward)"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mindtheedge_amd import _build  # noqa: E402

_build.build()
so, obj = "/tmp/libmte_i8stamps.so", "/tmp/conv_igemm8_stamps.o"
src = os.environ.get("IGEMM8_SRC", os.path.join(_build.CSRC, "conv_igemm8.hip"))      # (another version of the source / extra -D switches: same-box comparisons)
defs = [d for d in os.environ.get("IGEMM8_DEFS", "").split() if d]
subprocess.check_call([_build._hipcc()] + _build.FLAGS + ["-DMTE_STAMPS", "-DMTE_DEV", "-I", _build.CSRC, "-x", "hip"] + defs + ["-c", src, "-o", obj])
print("source %s %s" % (os.path.basename(src), " ".join(defs)))
others = [o for o in glob.glob(os.path.join(_build.CSRC, "dev", "*.o")) if not o.endswith("conv_igemm8.o")]
subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-o", so, obj] + others)
os.environ["MTE_LIB_PATH"] = so
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402

raw = ctypes.CDLL(so)
B = 8
K.use_patch_kernels(False)
raw.mte_debug_set(23, int(os.environ.get("IGEMM8_KNOB", "7")))
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(512, 512, 3, 24, 80), (256, 256, 3, 48, 160), (128, 128, 3, 96, 320), (32, 128, 7, 192, 640),
                                                                         (64, 256, 5, 96, 320), (256, 4096, 3, 24, 80)]


def timed(f):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


for shp in shapes:
    cin, cout, k, H, W = shp[:5]
    g = torch.Generator().manual_seed(1)
    x = K.new_act(B, cin, H, W)
    x.copy_(torch.randn(B, cin, H, W, generator=g).cuda())
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).cuda()
    b = torch.zeros(cout, device="cuda")
    pack = K.WeightPack()
    wf, _ = pack.get(w, x.dtype, True)
    saved = K._splitk_workspace
    K._splitk_workspace = lambda *a: (None, 0)
    us = timed(lambda: K.conv_forward(x, wf, b, cout, k, k))
    K._splitk_workspace = saved
    print("%d -> %d k%d @%dx%d: %.1f us per launch (stamped build)" % (cin, cout, k, H, W, us))
    torch.cuda.synchronize()
    n = 4096 * 24
    arr = (ctypes.c_ulonglong * n)()
    assert raw.mtei_igemm8_stamps(arr, n) == 0
    a = np.frombuffer(arr, dtype=np.uint64).reshape(-1, 24).astype(np.float64)
    a = a[a[:, 17] > 0]
    for grp in (0, 1):
        sel = a[(np.arange(len(a)) % 8 >= 4) == bool(grp)]
        per = sel[:, :17] / sel[:, 17:18]
        m = np.median(per, axis=0)
        print("%d -> %d k%d @%dx%d, wave group %d: %d waves, %d K-tiles; s_memtime ticks per K-tile (median): total %.0f" % (cin, cout, k, H, W, grp, len(sel), int(np.median(sel[:, 17])), m[16]))
        for ph in range(4):
            print("      phase %d: load section %6.1f | first barrier %6.1f | MFMAs %6.1f | second barrier %6.1f" % (ph + 1, m[4 * ph], m[4 * ph + 1], m[4 * ph + 2], m[4 * ph + 3]))
