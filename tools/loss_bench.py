#!/usr/bin/env python3
"""Loss-kernel microbenchmark (development aid): the fused 4-scale depth-edge + silog forward / backward at T8 size, replayed
from a HIP graph (the Python enqueue must not be what is measured) -> us per launch and algorithmic GB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("EDGE_DEFS"):                                   # private diagnostic build with extra -D switches (ablations)
    import glob
    import subprocess
    from mindtheedge_amd import _build
    _build.build()
    so, obj = "/tmp/libmte_edge_ab.so", "/tmp/edge_loss_ab.o"
    subprocess.check_call([_build._hipcc()] + _build.FLAGS + ["-DMTE_DEV", "-I", _build.CSRC] + os.environ["EDGE_DEFS"].split() +
                          ["-c", os.path.join(_build.CSRC, "edge_loss.hip"), "-o", obj])
    others = [o for o in glob.glob(os.path.join(_build.CSRC, "dev", "*.o")) if not o.endswith("edge_loss.o")]
    subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-o", so, obj] + others)
    os.environ["MTE_LIB_PATH"] = so
    print("private build:", os.environ["EDGE_DEFS"])
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402
from mindtheedge_amd.utils.synthetic import synthetic_batch  # noqa: E402


def main():
    B, H, W = 8, 384, 1280
    dev = torch.device("cuda")
    batch = synthetic_batch(B, H, W, 3, dev)
    sfx = ["", "_1", "_2", "_3"]
    invs = [(0.05 + 1.9 * torch.rand(B, 1, H >> s, W >> s, device=dev)).requires_grad_(True) for s in range(4)]
    edges = [batch["edge" + s] for s in sfx]
    normals = [batch["normal" + s] for s in sfx]
    px = sum(B * (H >> s) * (W >> s) for s in range(4))
    import ctypes
    lib = K.lib
    S = 4
    preds = [i.detach() for i in invs]
    dpreds = [torch.empty_like(p) for p in preds]
    arr = (K._EdgeScale * S)()
    for o, p, e, n, d in zip(arr, preds, edges, normals, dpreds):
        o.pred, o.edge, o.normal, o.mask, o.gmap, o.dpred = p.data_ptr(), e.data_ptr(), n.data_ptr(), None, None, d.data_ptr()
        o.H, o.W = p.shape[-2], p.shape[-1]
    iters = 20
    # as the training step hands it over: workspaces carved from a zeroed arena (MTE_OPT_LOSS_PREZEROED; one fill per `iters` launches here)
    prezeroed = os.environ.get("LOSS_PREZEROED", "1") == "1"
    lib.set_option(1, 1 if prezeroed else 0)
    nwork = (lib.mte_edge_loss_work_elems(ctypes.addressof(arr), S, B) + 31) // 32 * 32
    works = torch.zeros((iters, nwork), dtype=torch.float64, device=dev)
    slot = [0]
    losses = torch.empty((S + 1,), dtype=torch.float32, device=dev)
    coef = torch.empty((S * (2 * B + 1),), dtype=torch.float32, device=dev)
    aux = torch.empty((2,), dtype=torch.float32, device=dev)
    gl = torch.ones((S + 1,), dtype=torch.float32, device=dev)
    for with_silog in (False, True):
        gt = batch["depth"].data_ptr() if with_silog else 0

        def fwd():
            work = works[slot[0] % iters]
            slot[0] += 1
            lib.mte_edge_loss_multi_fwd(ctypes.addressof(arr), S, B, 1, 1, 1, 4.0, 10.0, 1.0, gt, work.data_ptr(), losses.data_ptr(),
                                        coef.data_ptr(), losses.data_ptr() + 16 if gt else 0, aux.data_ptr() if gt else 0, K._stream())

        def bwd():
            lib.mte_edge_loss_multi_bwd(ctypes.addressof(arr), S, B, 1, 1, 1, 4.0, coef.data_ptr(), gl.data_ptr(), gt,
                                        aux.data_ptr() if gt else 0, gl.data_ptr() + 16 if gt else 0, K._stream())
        res = {}
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            works.zero_(); slot[0] = 0
            fwd()
            torch.cuda.synchronize()

            def body():
                if prezeroed and name == "fwd":
                    works.zero_()
                slot[0] = 0
                for _ in range(iters):
                    fn()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                body()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                body()
            graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / iters * 1e3
        extra = B * H * W * 4 if with_silog else 0
        fb, bb = 12.0 * px + extra, 16.0 * px + extra
        f, b = res["fwd"], res["bwd"]
        print("silog fused=%d  fwd %6.1f us (%5.0f GB/s)  bwd %6.1f us (%5.0f GB/s)  both %6.1f us (%5.0f GB/s = %.3f of 8 TB/s)"
              % (with_silog, f, fb / f / 1e3, b, bb / b / 1e3, f + b, (fb + bb) / (f + b) / 1e3, (fb + bb) / (f + b) / 1e3 / 8000))
    print("losses", losses.tolist())


if __name__ == "__main__":
    main()
