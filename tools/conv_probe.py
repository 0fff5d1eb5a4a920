"""Development aid: run one conv shape (fwd, dgrad, wgrad) a few times -- for rocprofv3 --pmc / timing experiments."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MTE_USE_DEV_LIB", "1")      # development knobs live in libmte_hip_dev.so (-DMTE_DEV) only
import torch
from mindtheedge_amd import kernels as K

def main():
    B, H, W, cin, cout, k = [int(v) for v in sys.argv[1:7]]
    iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
    patch = (sys.argv[8] != "nopatch") if len(sys.argv) > 8 else True
    K.use_patch_kernels(patch)
    if os.environ.get('MTE_IGEMM_DMA') is not None:
        K.lib.mte_debug_set(0, int(os.environ['MTE_IGEMM_DMA']))
    for kv in filter(None, os.environ.get("MTE_DEBUG_KNOBS", "").split(",")):
        kk, vv = kv.split("=")
        K.lib.mte_debug_set(int(kk), int(vv))
    x = K.new_act(B, K.round8(cin), H, W); x.normal_()
    w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).requires_grad_(True)
    b = torch.zeros(cout, device="cuda", requires_grad=True)
    x.requires_grad_(True)
    pack = K.WeightPack()
    g = None
    for it in range(iters + 2):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        y = K.ConvFn.apply(x, w, b, pack)
        if g is None:
            g = torch.randn_like(y)
        y.backward(g)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    fl = 2.0 * B * H * W * cin * cout * k * k * 3
    print("shape %s: %.3f ms per fwd+bwd, %.1f TFLOP/s" % (sys.argv[1:7], dt * 1e3, fl / dt / 1e12))

main()
