#!/usr/bin/env python3
"""development aid: InvDepth head kernels (mte_invdepth_fwd / _bwd_data / _bwd_weight) at the four training shapes, time per launch
(HIP events over 20 launches) and achieved GB/s of algorithmic bytes; optional second library (argv[1]) for a same-box A/B."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mindtheedge_amd import _lib  # noqa: E402

libs = [("new", ctypes.CDLL(_lib.LIB_PATH))] + ([("old", ctypes.CDLL(sys.argv[1]))] if len(sys.argv) > 1 else [])
vp, cl, ci, cf = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float
for _, L in libs:
    L.mte_invdepth_fwd.argtypes = [vp, cl, vp, vp, vp, ci, ci, ci, ci, cf, ci, vp]
    L.mte_invdepth_bwd_data.argtypes = [vp, vp, vp, vp, vp, cl, ci, ci, ci, ci, cf, ci, vp]
    L.mte_invdepth_bwd_weight.argtypes = [vp, cl, vp, vp, ci, ci, ci, ci, ci, vp]
libs[0][1].mte_invdepth_bwd_weight.argtypes = [vp, cl, vp, vp, vp, ci, ci, ci, ci, ci, vp]
B = 8
st = torch.cuda.current_stream().cuda_stream
for C, H, W in ((32, 384, 1280), (64, 192, 640), (128, 96, 320), (256, 48, 160)):
    x = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16)
    w = torch.randn(C, 3, 3, device="cuda") * 0.1
    bias = torch.zeros(1, device="cuda")
    res = {}
    for name, L in libs:
        out = torch.empty(B, H, W, device="cuda")
        dout = torch.randn(B, H, W, device="cuda")
        dl = torch.empty(B, H, W, device="cuda")
        dx = torch.empty(B, H, W, C, device="cuda", dtype=torch.bfloat16)
        dwb = torch.empty(C * 9 + 1, device="cuda")
        rec = torch.empty(1024 * (C * 9 + 4), device="cuda")
        calls = {"fwd": lambda: L.mte_invdepth_fwd(x.data_ptr(), C, w.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W, C, 0.5, 0, st),
                 "bwd_data": lambda: L.mte_invdepth_bwd_data(w.data_ptr(), out.data_ptr(), dout.data_ptr(), dl.data_ptr(), dx.data_ptr(), C, B, H, W, C, 0.5, 0, st),
                 "bwd_weight": (lambda: L.mte_invdepth_bwd_weight(x.data_ptr(), C, dl.data_ptr(), dwb.data_ptr(), rec.data_ptr(), B, H, W, C, 0, st)) if name == "new"
                 else (lambda: L.mte_invdepth_bwd_weight(x.data_ptr(), C, dl.data_ptr(), dwb.data_ptr(), B, H, W, C, 0, st))}
        for k, f in calls.items():
            for _ in range(3):
                assert f() == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[(name, k)] = (e0.elapsed_time(e1) / 20, out.clone() if k == "fwd" else (dx.float().clone() if k == "bwd_data" else dwb.clone()))
    mb = B * H * W * C * 2 / 1e6
    for k in ("fwd", "bwd_data", "bwd_weight"):
        line = "C=%-3d %4dx%-4d %-10s" % (C, H, W, k)
        for name, _ in libs:
            ms = res[(name, k)][0]
            line += "  %s %7.1f us %5.2f TB/s" % (name, ms * 1e3, mb / ms / 1e3 / 1e3)
        if len(libs) == 2:
            a, b = res[("new", k)][1], res[("old", k)][1]
            line += "  max|new-old|/max|old| %.2e" % float((a - b).abs().max() / b.abs().max())
        print(line)
