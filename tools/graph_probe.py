"""development aid: capture the eval forward of PackNetSAN01 in a HIP graph (torch.cuda.CUDAGraph) and compare with eager."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mindtheedge_amd  # noqa: F401,E402
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402
from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K.set_compute_dtype("bf16")
torch.manual_seed(0)
net = PackNetSAN01(dropout=None, version="1A").cuda().eval()
static_rgb = torch.rand(B, 3, 384, 1280, device="cuda")


def fwd():
    with torch.no_grad():
        return net(static_rgb)["inv_depths"][0][0]


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


eager_ms = timed(fwd)
ref = fwd().float().clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        fwd()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_out = fwd()
torch.cuda.synchronize()
graph_ms = timed(g.replay)
new = torch.rand(B, 3, 384, 1280, device="cuda")
static_rgb.copy_(new)
g.replay()
torch.cuda.synchronize()
got = static_out.float().clone()
static_rgb.copy_(new)
want = fwd().float()
err = float((got - want).abs().max() / want.abs().max())
print({"B": B, "eager_ms": round(eager_ms, 3), "graph_ms": round(graph_ms, 3), "rel_err_vs_eager_on_new_input": err,
       "changed_with_input": bool((got - ref).abs().max() > 0)})
