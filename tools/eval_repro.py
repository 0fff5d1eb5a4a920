"""development diagnostic: N eval forwards of the benchmark network on one batch; count calls whose depth differs from the first (MTE_DEBUG_KNOBS=k=v,... applied first)"""
import sys, torch, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import infer_edges
from test_gpu_entry_points import _yaml
from mindtheedge_amd.models.model_wrapper import ModelWrapper
from mindtheedge_amd.utils.config import load_config
from mindtheedge_amd import kernels as K
import pathlib, tempfile
for kv in filter(None, os.environ.get("MTE_DEBUG_KNOBS", "").split(",")):
    k, v = kv.split("=")
    K.lib.mte_debug_set(int(k), int(v))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
tmp = pathlib.Path(tempfile.mkdtemp())
w = ModelWrapper(load_config(_yaml(tmp, 384, 1280))).cuda()
img = torch.rand(8, 3, 384, 1280, generator=torch.Generator().manual_seed(2)).cuda()
ref = infer_edges.infer_depth(w, img).clone()
bad = []
for i in range(1, N):
    d = infer_edges.infer_depth(w, img)
    if not torch.equal(d, ref):
        diff = (d - ref).abs()
        per = [bool((diff[b] > 0).any()) for b in range(8)]
        bad.append((i, float(diff.max()), per))
print("%s lib=%s knobs=%s: %d of %d calls differ from the first %s" % (sys.argv[2] if len(sys.argv) > 2 else "", os.environ.get("MTE_LIB_PATH", "tree"), os.environ.get("MTE_DEBUG_KNOBS", ""), len(bad), N - 1, bad[:4]))
