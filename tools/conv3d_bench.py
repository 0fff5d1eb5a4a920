#!/usr/bin/env python3
"""development aid (round 4): the conv3d(1 -> 4) data paths of the pack / unpack layers at their T8 shapes -- us per launch, algorithmic TB/s,
cycles per output and SIMD -- for the kernel variants behind mte_debug_set(1, v): 300 = fp32-VALU LDS stencils, 300 + bits = matrix-core forms.
Also prints the largest element-wise difference of each variant against the first one (same inputs).
usage: conv3d_bench.py [variant ...]   (default: 300 411 539)"""
import os
import sys

os.environ.setdefault("MTE_USE_DEV_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("P3_DEFS"):                                     # private diagnostic build of pack3d.hip with extra -D switches (ablations)
    import glob
    import subprocess
    from mindtheedge_amd import _build
    _build.build()
    so, obj = "/tmp/libmte_p3_ab.so", "/tmp/pack3d_ab.o"
    subprocess.check_call([_build._hipcc()] + _build.FLAGS + ["-DMTE_DEV", "-I", _build.CSRC] + os.environ["P3_DEFS"].split() +
                          ["-c", os.path.join(_build.CSRC, "pack3d.hip"), "-o", obj])
    others = [o for o in glob.glob(os.path.join(_build.CSRC, "dev", "*.o")) if not o.endswith("pack3d.o")]
    subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-o", so, obj] + others)
    os.environ["MTE_LIB_PATH"] = so
    print("private build:", os.environ["P3_DEFS"])
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402

variants = [int(v) for v in sys.argv[1:]] or [300, 411, 539]
B = 8
K.set_compute_dtype("bf16")
lib = K.lib
for extra in os.environ.get("P3_KNOBS", "").split():             # further values for mte_debug_set(1, .) applied once (e.g. 3001 / 3002 / 3004: output passes of the taps-in-K forward)
    lib.mte_debug_set(1, int(extra))
# (op, C, H, W): H, W of the UN-shuffled side (unpack: input of the layer; pack: input of the layer)
only = os.environ.get("P3_ONLY")
shapes = [("unpack_bwd_data", 32, 192, 640), ("unpack_bwd_data", 64, 96, 320), ("unpack_bwd_data", 128, 48, 160),
          ("unpack_fwd", 32, 192, 640), ("unpack_fwd", 64, 96, 320), ("unpack_fwd", 128, 48, 160), ("unpack_fwd", 256, 24, 80),
          ("pack_bwd_data", 32, 192, 640), ("pack_bwd_data", 64, 96, 320), ("pack_bwd_data", 128, 48, 160), ("pack_bwd_data", 256, 48, 160), ("pack_bwd_data", 512, 24, 80), ("pack_fwd", 256, 48, 160), ("pack_fwd", 512, 24, 80)]


def timed(f, n=10):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator().manual_seed(1)
w3 = ((torch.rand(4, 1, 3, 3, 3, generator=g) - 0.5) * 0.8).cuda()
b3 = ((torch.rand(4, generator=g) - 0.5) * 0.4).cuda()
for op, C, H, W in shapes:
    if only and only not in op:
        continue
    if op.startswith("unpack"):
        small, big = K.new_act(B, C, H, W), K.new_act(B, C, 2 * H, 2 * W)
        vol = B * H * W * C
    else:
        small, big = K.new_act(B, 16 * C, H // 2, W // 2), K.new_act(B, C, H, W)     # features [B,16C,H/2,W/2], un-packed side [B,C,H,W]
        vol = B * (H // 2) * (W // 2) * 4 * C
    small.copy_((torch.rand(small.shape, generator=g) * 2 - 1).cuda())
    big.copy_((torch.rand(big.shape, generator=g) * 2 - 1).cuda())
    sp, lds_ = K._pl(small)
    bp, ldb = K._pl(big)
    dt = K._dt(small)
    st = K._stream()
    if op == "unpack_bwd_data":      # dout = big -> dx = small
        call, out = (lambda: lib.mte_unpack3d_bwd_data(bp, ldb, w3.data_ptr(), sp, lds_, B, H, W, C, dt, st)), small
    elif op == "unpack_fwd":         # x = small -> out = big
        call, out = (lambda: lib.mte_unpack3d_fwd(sp, lds_, w3.data_ptr(), b3.data_ptr(), bp, ldb, B, H, W, C, dt, st)), big
    elif op == "pack_bwd_data":      # dout = small (features) -> dx = big
        call, out = (lambda: lib.mte_pack3d_bwd_data(sp, lds_, w3.data_ptr(), bp, ldb, B, H, W, C, dt, st)), big
    else:                            # x = big -> features = small
        call, out = (lambda: lib.mte_pack3d_fwd(bp, ldb, w3.data_ptr(), b3.data_ptr(), sp, lds_, B, H, W, C, dt, st)), small
    keep = out.float().clone()       # the op's INPUT may be `out` of another op: restore nothing, inputs are never written
    ref = None
    line = "%-16s C %3d @%3dx%-3d " % (op, C, H, W)
    for v in variants:
        lib.mte_debug_set(1, v)
        call()
        torch.cuda.synchronize()
        res = out.float().clone()
        us = timed(call)
        if ref is None:
            ref = res
            dtxt = ""
        else:
            d = (res - ref).abs()
            dtxt = "  max |d| %.2e (ref max %.2e)" % (float(d.max()), float(ref.abs().max()))
        line += "| v%d %7.1f us %5.2f TB/s %5.1f cyc/out/SIMD%s " % (v, us, 5 * vol * 2 / us / 1e6, us * 1e-6 * 2.4e9 * 1024 / vol, dtxt)
        out.copy_(keep) if False else None
    print(line)
lib.mte_debug_set(1, 539)
