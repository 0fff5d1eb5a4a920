#!/usr/bin/env python3
"""Run-to-run / eager-vs-HIP-graph determinism of the EVAL forward (round-3 item: GPUTEST_r02's red test).

For each frame size: N eager forwards and N graph replays of the same frame; every run is compared bit-for-bit with the first
eager run.  With --layers, forward hooks capture every layer's output of two eager runs and name the FIRST layer (execution
order) whose output differs -- i.e. the kernel that introduces the non-determinism.

    python tools/determinism_probe.py [--runs 20] [--sizes 96x160,384x1280] [--batch 1] [--layers]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=20)
    ap.add_argument("--sizes", type=str, default="96x160,384x1280")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--layers", action="store_true")
    args = ap.parse_args()
    import torch
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
    from mindtheedge_amd.utils.graph import GraphedDepth
    from mindtheedge_amd.networks.layers.packnet import layers01 as L

    torch.manual_seed(42)
    net = PackNetSAN01(dropout=0.5, version="1A").cuda().eval()
    leaf = (L.Conv2D, L.ResidualConv, L.PackLayerConv3d, L.UnpackLayerConv3d, L.InvDepth)
    for size in args.sizes.split(","):
        H, W = (int(v) for v in size.split("x"))
        rgb = torch.rand(args.batch, 3, H, W, generator=torch.Generator().manual_seed(0)).cuda()

        def fwd():
            with torch.no_grad():
                return [t.clone() for t in net(rgb)["inv_depths"][0]]

        ref = fwd()
        torch.cuda.synchronize()
        worst, ndiff_runs = 0.0, 0
        for _ in range(args.runs):
            out = fwd()
            d = max(float((a - b).abs().max()) for a, b in zip(out, ref))
            worst = max(worst, d)
            ndiff_runs += int(d != 0.0)
        print("[%s B%d] eager vs eager : %d / %d runs differ bitwise, max |d inv| = %.3e (inv-depth in (0,2))" % (size, args.batch, ndiff_runs, args.runs, worst))
        g = GraphedDepth(net, rgb)
        worst, ndiff_runs = 0.0, 0
        for _ in range(args.runs):
            out = [t.clone() for t in g(rgb)["inv_depths"][0]]
            d = max(float((a - b).abs().max()) for a, b in zip(out, ref))
            worst = max(worst, d)
            ndiff_runs += int(d != 0.0)
        print("[%s B%d] graph vs eager : %d / %d replays differ bitwise, max |d inv| = %.3e" % (size, args.batch, ndiff_runs, args.runs, worst))
        del g
        if args.layers:
            names = {m: n for n, m in net.named_modules()}
            caps = []

            def hook(mod, inp, out):
                caps[-1].append((names[mod], out.detach().float().clone()))

            hs = [m.register_forward_hook(hook) for m in net.modules() if isinstance(m, leaf)]
            first = {}
            for trial in range(args.runs):
                caps.append([])
                fwd()
                if trial == 0:
                    continue
                for (n0, a), (n1, b) in zip(caps[0], caps[-1]):
                    if not torch.equal(a, b):
                        first[n0] = first.get(n0, 0) + 1
                        break
                caps.pop()
            for h in hs:
                h.remove()
            order = [n for n, _ in caps[0]]
            print("[%s B%d] first layer whose output differs from run 0 (count over %d runs): %s" %
                  (size, args.batch, args.runs - 1, sorted(first.items(), key=lambda kv: order.index(kv[0])) or "none"))


if __name__ == "__main__":
    main()
