#!/usr/bin/env python3
"""development aid: where a wave of the implicit GEMM spends a K-step (s_memtime sums over the main loop: wait for the stage | barrier |
DMA issue | fragment reads + MFMA issue).  Private diagnostic build (conv_igemm.hip with -DMTE_STAMPS) under /tmp.
usage: igemm_stamps.py cin,cout,k,H,W[,ld] ...  (B = 8, forward)"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mindtheedge_amd import _build  # noqa: E402

so, obj = "/tmp/libmte_istamps.so", "/tmp/conv_igemm_stamps.o"
subprocess.check_call([_build._hipcc()] + _build.FLAGS + ["-DMTE_STAMPS", "-c", os.path.join(_build.CSRC, "conv_igemm.hip"), "-o", obj])
others = [o for o in glob.glob(os.path.join(_build.CSRC, "*.o")) if not o.endswith("conv_igemm.o")]
subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-o", so, obj] + others)
os.environ["MTE_LIB_PATH"] = so
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402

raw = ctypes.CDLL(so)
B = 8
K.use_patch_kernels(False)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(512, 512, 3, 24, 80), (256, 256, 3, 48, 160), (128, 128, 3, 96, 320), (64, 256, 5, 96, 320), (4096, 256, 3, 24, 80)]
prev = None
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
    n = 16384 * 8
    zero = (ctypes.c_ulonglong * n)()
    for _ in range(4):
        K.conv_forward(x, wf, b, cout, k, k, pack=pack, w=w)
    torch.cuda.synchronize()
    arr = (ctypes.c_ulonglong * n)()
    assert raw.mtei_igemm_stamps(arr, n) == 0
    a = np.frombuffer(arr, dtype=np.uint64).reshape(-1, 8).astype(np.float64)
    if prev is not None and np.array_equal(a, prev):
        print("%d -> %d k%d @%dx%d: not stamped (this launch took the ping-pong loop or a split-K / fp32 path)" % (cin, cout, k, H, W))
        continue
    prev = a.copy()
    a = a[a[:, 5] > 0]
    ks = a[:, 5]
    per = a[:, :5] / ks[:, None]
    m = np.median(per, axis=0)
    if np.median(a[:, 6]) == 1:
        print("%d -> %d k%d @%dx%d (ping-pong loop): %d waves stamped, %d K-steps; s_memtime cycles per K-step and wave (median): load phase (reads + DMA "
              "issue + waits) %.1f | first barrier %.1f | MFMA issue %.1f | wait + second barrier %.1f | total %.1f" % (cin, cout, k, H, W, len(a), int(np.median(ks)), m[2], m[1], m[3], m[0], m[4]))
        continue
    print("%d -> %d k%d @%dx%d: %d waves stamped, %d K-steps; s_memtime cycles per K-step and wave (median): "
          "stage wait %.1f | barrier %.1f | DMA issue %.1f | fragment reads + MFMA issue %.1f | total %.1f" % (cin, cout, k, H, W, len(a), int(np.median(ks)), m[0], m[1], m[2], m[3], m[4]))
