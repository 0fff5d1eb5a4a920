import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindtheedge_amd import kernels as K
K.set_compute_dtype("fp32")
dev = torch.device("cuda")
def run(name, build):
    x = torch.randn(1, 32, 16, 32, device=dev)
    fn, xin = build(x)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    K.begin_graph_capture()
    with torch.cuda.graph(g):
        out = fn()
    res = []
    for i in range(4):
        xin.copy_(torch.randn_like(xin))
        with torch.no_grad():
            ref = fn().clone()
        g.replay()
        got = out.clone()
        torch.cuda.synchronize()
        res.append("%.1e" % float((got.float() - ref.float()).abs().max() / ref.float().abs().max()))
    print(name, res)

def b_torch(x):
    w = torch.randn(32, 32, 3, 3, device=dev)
    return (lambda: torch.nn.functional.conv2d(x, w, padding=1).relu()), x
def b_conv(x):
    from mindtheedge_amd.networks.layers.packnet.layers01 import Conv2D
    m = Conv2D(32, 32, 3, 1).cuda().eval()
    return (lambda: m(K.as_act(x)).float() if False else m(x)), x
def b_gn(x):
    xa = K.as_act(x)
    gm, bt = torch.ones(32, device=dev), torch.zeros(32, device=dev)
    return (lambda: K._gn_forward(xa, None, None, gm, bt, 1e-5)[0]), xa
def b_convraw(x):
    xa = K.as_act(x)
    w = torch.randn(32, 32, 3, 3, device=dev)
    pack = K.WeightPack()
    def f():
        wf, _ = pack.get(w, xa.dtype, False)
        return K.conv_forward(xa, wf, None, 32, 3, 3, pack=pack, w=w)
    return f, xa
with torch.no_grad():
    for name, b in (("torch", b_torch), ("gn", b_gn), ("convraw", b_convraw), ("conv2d_module", b_conv)):
        try:
            run(name, b)
        except Exception as e:
            print(name, "EXC", type(e).__name__, str(e)[:200])
