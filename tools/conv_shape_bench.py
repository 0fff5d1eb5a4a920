#!/usr/bin/env python3
"""development aid: forward, data gradient and weight gradient of single conv shapes (HIP events over 20 calls each).
usage: conv_shape_bench.py cin,cout,k,H,W[,ldx] ...   (B = 8; ldx: pixel stride of the input in channels, e.g. 96 for the 72-channel concat)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402

for kv in filter(None, os.environ.get("MTE_DEBUG_KNOBS", "").split(",")):       # e.g. MTE_USE_DEV_LIB=1 MTE_DEBUG_KNOBS=11=0
    K.lib.mte_debug_set(*(int(v) for v in kv.split("=")))
B = 8
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(72, 32, 3, 384, 1280, 96), (3, 32, 5, 384, 1280), (128, 128, 3, 96, 320)]


def timed(f):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


for shp in shapes:
    cin, cout, k, H, W = shp[:5]
    cp = K.round8(cin)
    ld = shp[5] if len(shp) > 5 else cp
    g = torch.Generator().manual_seed(1)
    buf = K.new_act(B, ld, H, W)
    buf.copy_(torch.randn(B, ld, H, W, generator=g).cuda())
    x = K.channel_slice(buf, 0, cp)
    dy = K.new_act(B, cout, H, W)
    dy.copy_(torch.randn(B, cout, H, W, generator=g).cuda())
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).cuda()
    b = torch.zeros(cout, device="cuda")
    pack = K.WeightPack()
    wf, wb = pack.get(w, x.dtype, True)
    fl = 2.0 * B * H * W * cin * cout * k * k
    t_f = timed(lambda: K.conv_forward(x, wf, b, cout, k, k, pack=pack, w=w))
    t_d = timed(lambda: K.conv_backward(x, dy, w, pack, True, need_dw=False)) if cin > 3 else float("nan")
    t_w = timed(lambda: K._conv_wgrad(x, dy, w, False, None, None))
    print("%4d -> %-4d k%d @%dx%-4d ld %-3d  fwd %7.1f us %6.1f TF   dgrad %7.1f us %6.1f TF   wgrad(+unpack) %7.1f us %6.1f TF" % (
        cin, cout, k, H, W, ld, t_f * 1e3, fl / t_f / 1e9, t_d * 1e3, fl / t_d / 1e9, t_w * 1e3, fl / t_w / 1e9))
