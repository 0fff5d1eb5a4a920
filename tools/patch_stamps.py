#!/usr/bin/env python3
"""development aid: where a workgroup of the LDS-patch forward kernels spends its time (in-kernel s_memrealtime stamps).

Builds a PRIVATE diagnostic library (conv_patch.hip with -DMTE_STAMPS, every other object from the in-tree product build) under /tmp and runs
single conv shapes on it; per phase (prologue, tap loop and store/wait+barrier of each slice, epilogue) the median over all workgroups.
usage: patch_stamps.py [cin,cout,k,H,W[,ld]] ...   (B = 8);  MTE_DEBUG_KNOBS is NOT available here (product sources)"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mindtheedge_amd import _build  # noqa: E402

so = "/tmp/libmte_stamps.so"
obj = "/tmp/conv_patch_stamps.o"
first = sys.argv[1] if len(sys.argv) > 1 else ""
extra = ["-DMTE_PATCH_FWD1"] if first == "v1" else []
subprocess.check_call([_build._hipcc()] + _build.FLAGS + ["-DMTE_STAMPS"] + extra + ["-c", os.path.join(_build.CSRC, "conv_patch.hip"), "-o", obj])
others = [o for o in glob.glob(os.path.join(_build.CSRC, "*.o")) if not o.endswith("conv_patch.o")]
subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-o", so, obj] + others)
os.environ["MTE_LIB_PATH"] = so
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402

raw = ctypes.CDLL(so)
B = 8
args = [a for a in sys.argv[1:] if a != "v1"]
shapes = [tuple(int(v) for v in a.split(",")) for a in args] or [(72, 32, 3, 384, 1280, 96), (32, 32, 3, 384, 1280), (64, 64, 3, 192, 640), (104, 64, 3, 192, 640, 128), (32, 32, 7, 384, 1280)]
for shp in shapes:
    cin, cout, k, H, W = shp[:5]
    cp = K.round8(cin)
    ld = shp[5] if len(shp) > 5 else cp
    g = torch.Generator().manual_seed(1)
    buf = K.new_act(B, ld, H, W)
    buf.copy_(torch.randn(B, ld, H, W, generator=g).cuda())
    x = K.channel_slice(buf, 0, cp)
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).cuda()
    b = torch.zeros(cout, device="cuda")
    pack = K.WeightPack()
    wf, _ = pack.get(w, x.dtype, True)
    for _ in range(4):
        K.conv_forward(x, wf, b, cout, k, k, pack=pack, w=w)
    torch.cuda.synchronize()
    n = 16384 * 16
    arr = (ctypes.c_ulonglong * n)()
    assert raw.mtei_patch_stamps(arr, n) == 0
    ns = (cp + 31) // 32
    nst = min(2 + 2 * ns + 1, 16)                          # (the kernel keeps 16 stamps per workgroup: layers with more than six slices show the first ones)
    tall = cout <= 32 and (cp <= 32 or k <= 3) and H >= 16
    tiles = (W // 32) * ((H + 15) // 16 if tall else (H + 7) // 8) * B
    nb = min(tiles, 16384)
    a = np.frombuffer(arr, dtype=np.uint64).reshape(16384, 16).astype(np.int64)[:nb, :nst]
    d = np.diff(a, axis=1) * 10.0                          # ns (100 MHz counter)
    t0 = a[:, 0].min()
    print("%d -> %d k%d @%dx%d ld %d (%s form): %d workgroups, launch span %.1f us; phases of a workgroup, median ns:" % (
        cin, cout, k, H, W, ld, "first" if first == "v1" else "dispatched", nb, (a[:, -1].max() - t0) * 0.01))
    names = (["prologue"] + sum([["taps s%d" % s, "wait+sync s%d" % s] for s in range(ns)], []) + ["epilogue"])[:nst - 1]
    print("   " + "  ".join("%s %.0f" % (nm, np.median(d[:, i])) for i, nm in enumerate(names)))
    print("   workgroup total median %.0f ns" % (np.median(a[:, -1] - a[:, 0]) * 10))
