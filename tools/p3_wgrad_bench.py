"""development aid: conv3d(1->4) weight-gradient kernels of the pack / unpack layers at the training shapes (B = 8, 384 x 1280):
matrix-core version (mte_debug_set(1, 201)) against the fp32-VALU LDS version (200), time per launch and max relative difference."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MTE_USE_DEV_LIB", "1")      # development knobs live in libmte_hip_dev.so (-DMTE_DEV) only
import torch
from mindtheedge_amd import kernels as K

K.set_compute_dtype("bf16")
lib = K.lib
def timed(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = 8
for kind, C, H, W in (("pack", 64, 384, 1280), ("pack", 64, 192, 640), ("pack", 128, 96, 320), ("pack", 256, 48, 160), ("pack", 512, 24, 80),
                      ("unpack", 512, 12, 40), ("unpack", 256, 24, 80), ("unpack", 128, 48, 160), ("unpack", 64, 96, 320), ("unpack", 64, 192, 640)):
    x = K.new_act(B, C, H, W); x.normal_()
    if kind == "pack":
        g = K.new_act(B, 16 * C, H // 2, W // 2); fn_ = lib.mte_pack3d_bwd_weight
        mb = (x.numel() + g.numel()) * 2 / 1e6
    else:
        g = K.new_act(B, C, 2 * H, 2 * W); fn_ = lib.mte_unpack3d_bwd_weight
        mb = (x.numel() + g.numel()) * 2 / 1e6
    g.normal_()
    xp, ldx = K._pl(x); gp, ldg = K._pl(g)
    res = {}
    line = "%-6s C=%3d %4dx%4d  %6.1f MB:" % (kind, C, H, W, mb)
    for mode in (0, 1, 2):
        lib.mte_debug_set(1, 200 + min(mode, 1))
        lib.mte_debug_set(1, 100 + (0 if mode == 1 else 1))           # mode 1: the larger tile tables
        out = torch.zeros(112, device="cuda")
        run = lambda: fn_(xp, ldx, gp, ldg, out.data_ptr(), B, H, W, C, K._dt(x), K._stream())
        us = timed(run)
        run(); torch.cuda.synchronize()
        res[mode] = out.clone()
        line += "  %s %7.1f us (%.2f TB/s)" % (("valu", "mfma bigtile", "mfma")[mode], us, mb / us)
    lib.mte_debug_set(1, 201); lib.mte_debug_set(1, 101)
    d = (res[0] - res[1]).abs().max().item() / res[0].abs().max().item()
    print(line + "  maxrel %.2e" % d)
