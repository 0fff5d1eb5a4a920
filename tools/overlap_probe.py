#!/usr/bin/env python3
"""development probe (round 5): what does a memory-bound main-queue kernel (GroupNorm forward / backward) lose while an MFMA weight-gradient kernel runs on the
side queue -- occupancy (the side kernel's workgroups hold CUs: 512 threads, 96 KB LDS, the whole register file of their SIMDs) or traffic?
Side stream: (a) nothing, (b) tools/probe/cu_hog.hip on 64 / 128 / 192 CUs, asleep (occupancy only) or issuing MFMAs back to back (no memory traffic either way), (c) the real nine-tap weight gradient
(4096 -> 256 @24x80) at its shared-chip width.  Main stream: GroupNorm statistics + apply / backward of the T8 layer classes, HIP events over 20 calls."""
import ctypes, os, subprocess, sys
os.environ.setdefault("MTE_USE_DEV_LIB", "1")      # GroupNorm geometry knobs as arguments: 3=4096 (workgroups aimed for) 2=16 (minimum pixel rows per thread)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindtheedge_amd import kernels as K

here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libcu_hog.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "probe", "cu_hog.hip")])
hog = ctypes.CDLL(so)
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
K.set_compute_dtype("bf16")
for kv in sys.argv[1:]:
    K.lib.mte_debug_set(*(int(v) for v in kv.split("=")))
print("knobs:", sys.argv[1:])
B = 8
side = torch.cuda.Stream()
out = torch.zeros(4096, device="cuda")
# the real side kernel
xw = K.new_act(B, 4096, 24, 80).normal_(); dyw = K.new_act(B, 256, 24, 80).normal_(); ww = torch.randn(256, 4096, 3, 3, device="cuda") * 0.01
x2 = K.new_act(B, 64, 192, 640).normal_(); dy2 = K.new_act(B, 64, 192, 640).normal_(); w2 = torch.randn(64, 64, 3, 3, device="cuda") * 0.01


def side_work(kind, n):
    with torch.cuda.stream(side):
        for _ in range(n):
            if kind[0] in ("hog", "sleep"):
                hog.hog_launch(kind[1], 96 * 1024, 400 if kind[0] == "hog" else 1600, out.data_ptr(), side.cuda_stream, int(kind[0] == "sleep"))          # ~0.3 ms per launch
            elif kind[0] == "w9":
                K._conv_wgrad(xw, dyw, ww, False, None, None)
            elif kind[0] == "patch":
                K._conv_wgrad(x2, dy2, w2, False, None, None)


CLASSES = [("128@96x320", 128, 96 * 320), ("64@192x640", 64, 192 * 640), ("32@384x1280", 32, 384 * 1280)]
gm = {C: torch.ones(C, device="cuda") for _, C, _ in CLASSES}; bt = {C: torch.zeros(C, device="cuda") for _, C, _ in CLASSES}
print("%-14s %-22s %10s %10s" % ("layer class", "side queue", "fwd us", "bwd us"))
for label, C, HW in CLASSES:
    y = K.new_act(B, C, HW, 1).normal_(); dz = K.new_act(B, C, HW, 1).normal_()
    stats = K._gn_forward(y, None, None, gm[C], bt[C], 1e-5)[1]
    for kind in (("none",), ("sleep", 64), ("sleep", 128), ("sleep", 192), ("hog", 64), ("hog", 128), ("hog", 192), ("w9",), ("patch",)):
        res = []
        for name in ("fwd", "bwd"):
            f = (lambda: K._gn_forward(y, None, None, gm[C], bt[C], 1e-5)) if name == "fwd" else \
                (lambda: K._gn_backward(dz, y, None, None, stats, gm[C], bt[C], 1e-5, False, want_dbias=True))
            for _ in range(3): f()
            torch.cuda.synchronize()
            if kind[0] != "none": side_work(kind, 40 if kind[0] in ("hog", "sleep") else 60)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record()
            e1.synchronize()
            busy = not side.query()
            torch.cuda.synchronize()
            res.append("%8.1f%s" % (e0.elapsed_time(e1) / 20 * 1e3, " " if busy or kind[0] == "none" else "*"))
        print("%-14s %-22s %10s %10s" % (label, " ".join(str(k) for k in kind), res[0], res[1]))
print("(* = the side queue ran dry before the 20 calls ended)")
