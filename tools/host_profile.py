"""development aid: where the HOST time of a training step goes (cProfile over a few eager steps; the GPU runs behind)."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mindtheedge_amd import kernels as K
from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
from mindtheedge_amd.models.SemiSupEdgeModel import SemiSupEdgeModel
from mindtheedge_amd.losses.grad_loss import GradLoss
from mindtheedge_amd.trainers.data_parallel import FlatParameters, FusedAdam
from mindtheedge_amd.utils.synthetic import synthetic_batch

K.set_compute_dtype("bf16")
torch.manual_seed(42)
dev = torch.device("cuda", 0)
net = PackNetSAN01(dropout=0.5, version="1A").to(dev)
model = SemiSupEdgeModel(supervised_method="sparse-silog", supervised_num_scales=1, supervised_loss_weight=1.0,
                         edges_depth_edge_loss_all_scales=True).to(dev)
model.add_depth_net(net)
model.add_edge_loss(GradLoss("cross_entropy", True, [], 10.0, 1.0))
model.train()
batch = synthetic_batch(8, 384, 1280, 0, dev)
flat = FlatParameters(net.parameters())
opt = FusedAdam(flat, lr=1e-4)
def step():
    opt.zero_grad()
    out = model(batch)
    out["loss"].backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(4): step()
pr.disable()
host = (time.perf_counter() - t0) / 4
torch.cuda.synchronize()
print("host ms/step under cProfile: %.1f" % (host * 1e3))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): step()
print("host ms/step, 3 steps enqueued after a sync, no profiler: %.1f" % ((time.perf_counter() - t0) / 3 * 1e3))
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(38)
print(s.getvalue()[:9000])

# ---- the backward half runs in the autograd engine's device thread: switch a profiler on from inside the first backward call
bpr = cProfile.Profile()
state = {"on": False}
orig = K.DepthLossesFn.backward
def first_backward(ctx, *a):
    if not state["on"]:
        bpr.enable(); state["on"] = True
    return orig(ctx, *a)
K.DepthLossesFn.backward = staticmethod(first_backward)
for _ in range(4): step()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(bpr, stream=s).sort_stats("tottime").print_stats(30)
print("==== autograd thread (4 steps)")
print(s.getvalue()[:7000])
