import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindtheedge_amd import kernels as K
from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
K.set_compute_dtype("fp32")
torch.manual_seed(0)
net = PackNetSAN01(dropout=None, version="1A").cuda().eval()
enc = net.encoder
g = torch.Generator().manual_seed(3)
fs = [torch.rand(1, 3, 64, 128, generator=g).cuda() for _ in range(3)]
def fwd(rgb):
    x = enc.pre_calc(rgb)
    c1 = enc.conv1(x)
    p3 = K.Pack3dFn.apply(c1, enc.pack1.conv3d.weight, enc.pack1.conv3d.bias)
    y = enc.pack1.conv(p3)
    return [x, c1, p3, y]
with torch.no_grad():
    refs = [[t.float().clone() for t in fwd(f)] for f in fs]
    rgb = fs[0].clone()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fwd(rgb)
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    K.begin_graph_capture()
    with torch.cuda.graph(gr):
        out = fwd(rgb)
    K.end_graph_capture()
def err(x, y): return "%.0e" % float((x.float() - y.float()).abs().max() / y.float().abs().max())
for i, f in enumerate(fs):
    rgb.copy_(f)
    gr.replay()
    torch.cuda.synchronize()
    print(i, "x, conv1, pack3d, pack1.conv:", [err(a, b) for a, b in zip(out, refs[i])])
