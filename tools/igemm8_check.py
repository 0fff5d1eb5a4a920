#!/usr/bin/env python3
"""development aid (round 4): the 8-phase implicit-GEMM kernels (csrc/conv_igemm8.hip) against the older tile forms.
  check : bit-equality with the 128x128 register... (knob 6 = 0, 23 = 0) result on awkward shapes (odd K-step counts, column tails, row tails,
          split-K, accumulate, channel-slice strides), repeated -- a stale LDS slot or an early fragment read shows on some repetitions only
  bench : the training step's forward / data-gradient shapes, old dispatch (23 = 0) vs new (23 = 3), interleaved rounds, HIP events
usage: igemm8_check.py [check|bench|both] [reps]      (development library: kernel-variant knobs)"""
import os
import sys

os.environ.setdefault("MTE_USE_DEV_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
K.use_patch_kernels(False)
S = K.lib.mte_debug_set


def make(cin, cout, k, B, H, W, ldx=None, seed=0):
    g = torch.Generator().manual_seed(3 + cin + cout + seed)
    cp = K.round8(cin)
    ld = ldx or cp
    w = ((torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * (3.0 / (cin * k * k)) ** 0.5).cuda()
    b = (torch.rand(cout, generator=g) - 0.5).cuda()
    buf = K.new_act(B, ld, H, W)
    buf.copy_((torch.rand(B, ld, H, W, generator=g) * 2 - 1).cuda())
    x = K.channel_slice(buf, 0, cp) if ld != cp else buf
    pack = K.WeightPack()
    wf, wb = pack.get(w, x.dtype, True)
    return x, wf, b, buf


def check():
    bad = 0
    # (cin, cout, k, B, H, W, ldx)
    shapes = [(64, 256, 3, 2, 24, 40, None), (96, 256, 3, 3, 10, 52, None), (32, 256, 3, 1, 30, 33, None), (32, 384, 5, 1, 17, 33, None),
              (128, 384, 1, 2, 16, 48, None), (256, 256, 3, 8, 48, 160, None), (64, 256, 5, 4, 96, 320, None), (384, 256, 3, 2, 48, 160, None),
              (128, 512, 3, 8, 24, 80, None), (256, 256, 1, 8, 48, 160, None), (64, 200, 3, 2, 40, 64, 96), (128, 128, 3, 2, 96, 320, None),
              (96, 128, 3, 1, 33, 47, None), (512, 104, 3, 1, 24, 80, None), (2048, 256, 3, 2, 12, 40, None), (1024, 512, 3, 1, 12, 40, None),
              (160, 136, 3, 1, 31, 45, 200), (32, 128, 7, 1, 64, 96, None)]
    for cin, cout, k, B, H, W, ldx in shapes:
        x, wf, b, _ = make(cin, cout, k, B, H, W, ldx)
        for accumulate in (False, True):
            y0 = K.new_act(B, K.round8(cout) + 8, H, W)          # output as a channel slice of a wider buffer
            y0.copy_(torch.randn(B, y0.shape[1], H, W).cuda())

            def run(v8, split):
                S(23, v8); S(24, 1000000 if split else 1); S(6, 0)
                saved = K._splitk_workspace
                if not split:
                    K._splitk_workspace = lambda *a: (None, 0)
                else:
                    K._splitk_workspace = lambda M, N, dev: (torch.empty((8 * M * N,), dtype=torch.float32, device=dev), 8 * M * N)
                out = y0.clone()
                ys = K.channel_slice(out, 0, K.round8(cout))
                K.conv_forward(x, wf, b, cout, k, k, out=ys, accumulate=accumulate)
                K._splitk_workspace = saved
                torch.cuda.synchronize()
                return out

            for split in (False, True):
              for v8, name in ((39, "tap-major"), (7, "slice-major")):
                ref = run(0, split)
                if v8 == 7 and not split:                        # (round 5) another order of the fp32 sum: one bf16 rounding from the 128 x 128 tile, bitwise from run to run
                    first = run(v8, False)
                    d = (first.float() - ref.float()).abs()
                    if int((d > ref.float().abs() * 2.0 ** -6 + 1e-3).sum()):
                        print("   slice-major result off by more than one bf16 ulp, max |d| %.3e" % float(d.max()))
                        bad += 1
                    ref = first
                if split:                                        # different split counts: equal up to the last bf16 rounding; the new kernel must repeat itself
                    first = run(v8, True)
                    d = (first.float() - ref.float()).abs()
                    tol = ref.float().abs() * 2.0 ** -6 + 1e-3       # (accumulate: the unsplit tile forms round conv + bias to bf16 before the sum, the split-K finish does not)
                    nbad = int((d > tol).sum())
                    if nbad:
                        print("   split-K result off: %d elements beyond one bf16 ulp, max |d| %.3e" % (nbad, float(d.max())))
                        bad += 1
                    ref = first
                miss = 0
                for r in range(reps):
                    y = run(v8, split)
                    if not torch.equal(y, ref):
                        miss += 1
                        if miss == 1:
                            d = (y.float() - ref.float()).abs()
                            print("   first mismatch rep %d: %d elements differ, max |d| %.3e (ref max %.3e)" % (r, int((d > 0).sum()), float(d.max()), float(ref.float().abs().max())))
                bad += miss
                print("%5d -> %-4d k%d B%d %3dx%-4d ldx %-4s acc %d split %d %-18s: %d / %d repetitions differ" % (cin, cout, k, B, H, W, ldx, accumulate, split, name, miss, reps))
    S(23, 51); S(24, 200); S(6, 3)
    print("MISMATCHES:", bad)
    return bad


def bench():
    B = 8
    # (cin_p, n, k, H, W, count per training step)  -- forward and data-gradient launches of the T8 step that have N >= 104 and Cin_p % 32 == 0
    shapes = [(32, 128, 7, 192, 640, 1), (4096, 256, 3, 24, 80, 1), (8192, 512, 3, 12, 40, 1), (256, 4096, 3, 24, 80, 1), (512, 8192, 3, 12, 40, 1),
              (512, 128, 5, 48, 160, 1), (64, 256, 5, 96, 320, 1), (128, 512, 5, 48, 160, 1), (64, 104, 3, 192, 640, 1), (512, 768, 3, 24, 80, 1),
              (128, 200, 3, 96, 320, 1), (768, 512, 3, 24, 80, 1), (256, 384, 3, 48, 160, 1), (128, 128, 3, 96, 320, 10), (512, 512, 3, 24, 80, 14),
              (384, 256, 3, 48, 160, 1), (256, 256, 3, 48, 160, 10), (512, 256, 3, 24, 80, 2), (64, 128, 3, 96, 320, 2), (256, 128, 3, 48, 160, 2),
              (128, 256, 3, 48, 160, 2), (256, 512, 3, 24, 80, 2), (512, 512, 3, 12, 40, 2), (128, 128, 1, 96, 320, 2), (256, 256, 1, 48, 160, 4),
              (512, 512, 1, 24, 80, 4),
              (192, 128, 3, 96, 320, 1), (128, 192, 3, 96, 320, 1)]       # iconv3 after the rank-1 split of its inverse-depth channel (forward, data gradient)
    tot = {0: 0.0, 3: 0.0, 7: 0.0, 35: 0.0}
    for cin, cout, k, H, W, cnt in shapes:
        x, wf, b, _ = make(cin, cout, k, B, H, W)
        fl = 2.0 * B * H * W * cin * cout * k * k
        t = {0: [], 3: [], 7: [], 35: []}
        for rnd in range(5):
            for v8 in (0, 3, 7, 35):
                S(23, v8)
                for _ in range(2):
                    K.conv_forward(x, wf, b, cout, k, k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    K.conv_forward(x, wf, b, cout, k, k)
                e1.record()
                torch.cuda.synchronize()
                t[v8].append(e0.elapsed_time(e1) / 10)
        m = {v: sorted(t[v])[len(t[v]) // 2] for v in t}
        for v in m:
            tot[v] += m[v] * cnt
        print("%5d -> %-4d k%d @%3dx%-4d x%-2d  old %7.1f us %6.0f TF   dispatch rule %7.1f us %6.0f TF %+5.1f %%   8-phase everywhere %7.1f us %6.0f TF %+5.1f %%   rule, tap-major %7.1f us %6.0f TF %+5.1f %%" % (
            cin, cout, k, H, W, cnt, m[0] * 1e3, fl / m[0] / 1e9, m[3] * 1e3, fl / m[3] / 1e9, (m[0] / m[3] - 1) * 100, m[7] * 1e3, fl / m[7] / 1e9, (m[0] / m[7] - 1) * 100,
            m[35] * 1e3, fl / m[35] / 1e9, (m[0] / m[35] - 1) * 100))
    print("weighted sum per step: old %.3f ms, dispatch rule (slice-major) %.3f ms, 8-phase everywhere %.3f ms, dispatch rule tap-major %.3f ms" % (tot[0], tot[3], tot[7], tot[35]))
    S(23, 51)


rc = 0
if mode in ("check", "both"):
    rc = check()
if mode in ("bench", "both"):
    bench()
sys.exit(1 if rc else 0)
