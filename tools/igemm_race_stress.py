#!/usr/bin/env python3
"""development aid: race screen for the implicit-GEMM tile variants.  Every variant of a shape must reproduce the 4-wave 128x128 result
BIT FOR BIT (same K order, same fp32 accumulation chain per output), run after run: a stale LDS slot or a fragment read that overtakes its
DMA shows up as a mismatch on some repetition.  usage: igemm_race_stress.py [repetitions [noise]]   (development library: kernel-variant knobs)"""
import os
import sys

os.environ.setdefault("MTE_USE_DEV_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
K.use_patch_kernels(False)
K._splitk_workspace = lambda *a: (None, 0)
shapes = [(64, 128, 3, 2, 24, 40), (96, 256, 3, 3, 10, 52), (32, 128, 5, 1, 30, 33), (128, 384, 1, 2, 16, 48),
          (256, 256, 3, 8, 48, 160), (64, 256, 5, 4, 96, 320), (384, 256, 3, 2, 48, 160), (128, 512, 3, 8, 24, 80), (256, 256, 1, 8, 48, 160)]
noise = torch.cuda.Stream() if len(sys.argv) > 2 else None
na = torch.empty(256 << 20, dtype=torch.uint8, device="cuda"); nb = torch.empty_like(na)
nm = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16); nmo = torch.empty_like(nm)
bad = 0
for cin, cout, k, B, H, W in shapes:
    g = torch.Generator().manual_seed(3 + cin + cout)
    w = ((torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * (3.0 / (cin * k * k)) ** 0.5).cuda()
    b = (torch.rand(cout, generator=g) - 0.5).cuda()
    xa = K.image_to_act(torch.rand(B, cin, H, W, generator=g).cuda() * 2 - 1)
    pack = K.WeightPack()
    wf, wb = pack.get(w, xa.dtype, True)

    def run(big, pp, v8=0):
        K.lib.mte_debug_set(6, big)
        K.lib.mte_debug_set(7, 1)
        K.lib.mte_debug_set(21, pp)
        K.lib.mte_debug_set(23, v8)                   # 8-phase kernels (round 4): 0 off, 7 every eligible launch (round 5: slice-major K order), 39 = 7 | 32 the same tap-major
        K.lib.mte_debug_set(24, 1)
        if noise is not None:                         # something else on the chip: another queue streaming HBM and a GEMM under the launch
            with torch.cuda.stream(noise):
                nb.copy_(na)
                torch.mm(nm, nm, out=nmo)
        y = K.conv_forward(xa, wf, b, cout, k, k)
        torch.cuda.synchronize()
        return y

    ref = run(0, 0).clone()
    for big, pp, v8, name in ((1, 0, 0, "256x128 8 waves"), (2, 0, 0, "256x256 16 waves"), (2, 1, 0, "256x256 ping-pong"),
                              (0, 0, 39, "8-phase tap-major"), (0, 0, 7, "8-phase slice-major")):
        miss = 0
        own = run(big, pp, v8).clone() if v8 == 7 else ref     # (slice-major: another summation order -- it must repeat ITSELF bit for bit)
        for r in range(reps):
            y = run(big, pp, v8)
            if not torch.equal(y, own):
                miss += 1
                if miss == 1:
                    d = (y.float() - own.float()).abs()
                    print("   first mismatch at rep %d: %d elements differ, max |d| %.3e" % (r, int((d > 0).sum()), float(d.max())))
        bad += miss
        print("%4d -> %-4d k%d B%d %dx%-4d %-20s %d / %d repetitions differ" % (cin, cout, k, B, H, W, name, miss, reps))
K.lib.mte_debug_set(6, 3); K.lib.mte_debug_set(7, 224); K.lib.mte_debug_set(21, 1); K.lib.mte_debug_set(23, 51); K.lib.mte_debug_set(24, 200)
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
