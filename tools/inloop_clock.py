#!/usr/bin/env python3
"""development aid (round 6): the shader clock the chip HOLDS inside the MFMA loops (MI355X_MICROARCH.md 'DVFS give-back' item 6).

Private diagnostic build (-DMTE_CLOCK: conv_patch.hip, conv_igemm8.hip, conv_wgrad9.hip; every other object from the in-tree development build) under /tmp.
Each kernel stamps s_memtime and s_memrealtime ONCE in front of and once behind its main loop; clock = delta s_memtime / delta s_memrealtime x 100 MHz, median over
the workgroups of the LAST launch after >= SECONDS of back-to-back launches on random data, (i) alone, (ii) beside tools/probe/cu_hog.hip issuing MFMAs on 64 CUs
on a second queue, (iii) beside the GroupNorm backward of the 64-channel 192x640 layer class on a second queue.
usage: inloop_clock.py [seconds]   -> profiles/r06_inloop_clock.txt"""
import ctypes
import glob
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mindtheedge_amd import _build  # noqa: E402

_build.build()
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
so = "/tmp/libmte_clock.so"
objs = []
for f in ("conv_patch", "conv_igemm8", "conv_wgrad9"):
    o = "/tmp/%s_clock.o" % f
    subprocess.check_call([_build._hipcc()] + _build.FLAGS + ["-DMTE_CLOCK", "-DMTE_DEV", "-c", os.path.join(_build.CSRC, f + ".hip"), "-o", o])
    objs.append(o)
others = [o for o in glob.glob(os.path.join(_build.CSRC, "dev", "*.o")) if os.path.basename(o)[:-2] not in ("conv_patch", "conv_igemm8", "conv_wgrad9")]
subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-o", so] + objs + others)
os.environ["MTE_LIB_PATH"] = so
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402

raw = ctypes.CDLL(so)
hog_so = "/tmp/libcu_hog.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", hog_so, os.path.join(ROOT, "tools", "probe", "cu_hog.hip")])
hog = ctypes.CDLL(hog_so)
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
K.set_compute_dtype("bf16")
B = 8
side = torch.cuda.Stream()
hog_out = torch.zeros(4096, device="cuda")


def read_clock(tag):
    n = 16384 * 2
    arr = (ctypes.c_ulonglong * n)()
    assert getattr(raw, "mtei_clk_" + tag)(arr, n) == 0
    a = np.frombuffer(arr, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
    a = a[a[:, 1] > 0]
    if not len(a):
        return None
    ghz = a[:, 0] / a[:, 1] * 0.1
    return len(a), float(np.median(ghz)), float(np.percentile(ghz, 10)), float(np.percentile(ghz, 90)), float(np.median(a[:, 1]) * 0.01)


def conv_case(cin, cout, k, H, W, what, ld=None):
    cp = K.round8(cin)
    ld = ld or cp
    g = torch.Generator().manual_seed(1)
    buf = K.new_act(B, ld, H, W)
    buf.copy_(torch.randn(B, ld, H, W, generator=g).cuda())
    x = K.channel_slice(buf, 0, cp)
    dy = K.new_act(B, cout, H, W)
    dy.copy_(torch.randn(B, cout, H, W, generator=g).cuda())
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).cuda()
    b = torch.zeros(cout, device="cuda")
    pack = K.WeightPack()
    wf, _ = pack.get(w, x.dtype, True)
    if what == "fwd":
        return lambda: K.conv_forward(x, wf, b, cout, k, k, pack=pack, w=w)
    if what == "dgrad":
        return lambda: K.conv_backward(x, dy, w, pack, True, need_dw=False)
    return lambda: K._conv_wgrad(x, dy, w, False, None, None)


# GroupNorm backward of the 64-channel 192x640 class (the memory-bound neighbour of the real step)
_gC, _gHW = 64, 192 * 640
_gy = K.new_act(B, _gC, _gHW, 1).normal_()
_gdz = K.new_act(B, _gC, _gHW, 1).normal_()
_ggm, _gbt = torch.ones(_gC, device="cuda"), torch.zeros(_gC, device="cuda")
_gstats = K._gn_forward(_gy, None, None, _ggm, _gbt, 1e-5)[1]


def side_launch(kind):
    with torch.cuda.stream(side):
        if kind == "hog64":
            hog.hog_launch(64, 96 * 1024, 4000, hog_out.data_ptr(), side.cuda_stream, 0)           # ~3 ms of back-to-back MFMAs on 64 CUs
        elif kind == "gn_bwd":
            K._gn_backward(_gdz, _gy, None, None, _gstats, _ggm, _gbt, 1e-5, False, want_dbias=True)


def run(f, tag, kind):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    read_clock(tag)                                                   # (read-and-clear: only this case's launches are in the buffer afterwards)
    t_end = time.time() + SECONDS
    ring = []
    n = 0
    while time.time() < t_end:
        if kind != "alone" and (side.query() or n % (40 if kind == "hog64" else 3) == 0):
            side_launch(kind)
        f()
        n += 1
        ev = torch.cuda.Event()
        ev.record()
        ring.append(ev)
        if len(ring) > 64:
            ring.pop(0).synchronize()                                 # bounded run-ahead, the queue never drains
    e0.record()                                                       # us / launch at the END of the run (warm clocks), with the neighbour still queued
    for _ in range(10):
        if kind != "alone" and side.query():
            side_launch(kind)
        f()
    e1.record()
    busy = kind == "alone" or not side.query()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    r = read_clock(tag)
    return us, n, r, busy


CASES = [
    ("conv_igemm8_kernel           256 -> 256 k3 @48x160 fwd", "igemm8", lambda: conv_case(256, 256, 3, 48, 160, "fwd")),
    ("conv_igemm8_kernel           512 -> 512 k3 @24x80  fwd", "igemm8", lambda: conv_case(512, 512, 3, 24, 80, "fwd")),
    ("conv_patch_fwd2_kernel<7,1>  32 -> 32  k7 @384x1280 fwd, 32x32x16", "patch", lambda: conv_case(32, 32, 7, 384, 1280, "fwd"), (11, 500)),
    ("conv_patch_fwd2_kernel<7,1>  32 -> 32  k7 @384x1280 fwd, 16x16x32", "patch", lambda: conv_case(32, 32, 7, 384, 1280, "fwd"), (11, 501)),
    ("conv_patch_fwd2_kernel<3,1,T> 64 -> 32 k3 @192x640 fwd", "patch", lambda: conv_case(64, 32, 3, 192, 640, "fwd")),
    ("conv_patch_fwd_kernel<3,2>   64 -> 64  k3 @192x640 fwd", "patch", lambda: conv_case(64, 64, 3, 192, 640, "fwd")),
    ("conv_patch_fwd2_kernel<5,2>  256 -> 64 k5 @96x320  fwd, 32x32x16", "patch", lambda: conv_case(256, 64, 5, 96, 320, "fwd"), (11, 500)),
    ("conv_patch_fwd2_kernel<5,2>  256 -> 64 k5 @96x320  fwd, 16x16x32", "patch", lambda: conv_case(256, 64, 5, 96, 320, "fwd"), (11, 501)),
    ("conv_wgrad9_kernel           256 -> 256 k3 @48x160 wgrad", "wgrad9", lambda: conv_case(256, 256, 3, 48, 160, "wgrad")),
    ("conv_patch_wgrad_kernel<3,2> 64 -> 64  k3 @192x640 wgrad", "patch", lambda: conv_case(64, 64, 3, 192, 640, "wgrad")),
    ("conv_patch_wgrad_kernel<7,1> 32 -> 32  k7 @384x1280 wgrad", "patch", lambda: conv_case(32, 32, 7, 384, 1280, "wgrad")),
]
print("in-loop shader clock (delta s_memtime / delta s_memrealtime x 100 MHz around the main loop; median [p10 .. p90] over the workgroups of the last launch");
print("after >= %.1f s of back-to-back launches, random data, B = 8).  loop us = median time a workgroup spends between the two stamps." % SECONDS)
print("%-68s %-8s %9s %7s %6s  %s" % ("kernel / shape", "beside", "us/launch", "launches", "WGs", "GHz median [p10 .. p90]   loop us"))
for case in CASES:
    name, tag, mk = case[:3]
    if len(case) > 3:
        K.lib.mte_debug_set(*case[3])
    f = mk()
    for kind in ("alone", "hog64", "gn_bwd"):
        us, n, r, busy = run(f, tag, kind)
        if r is None:
            print("%-68s %-8s %9.1f %7d   (no stamps: another kernel ran)" % (name, kind, us, n))
            continue
        print("%-68s %-8s %9.1f %7d %6d  %.3f [%.3f .. %.3f]   %6.2f%s" % (name, kind, us, n, r[0], r[1], r[2], r[3], r[4], "" if busy else "  (side queue ran dry)"))
    del f
    torch.cuda.empty_cache()
