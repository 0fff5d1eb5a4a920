#!/usr/bin/env python3
"""GroupNorm+ELU kernel microbenchmark (development aid): forward (statistics + apply, or the single-pass slab kernel) and
backward of every layer class of the T8 step, timed serially with HIP events over buffer sets that rotate through more
than the Infinity Cache, printed as us per call and algorithmic GB/s.  `python tools/gn_bench.py [knob=value ...]`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MTE_USE_DEV_LIB", "1")      # development knobs live in libmte_hip_dev.so (-DMTE_DEV) only
import torch  # noqa: E402
from mindtheedge_amd import kernels as K  # noqa: E402

CLASSES = [("tiny 512@12x40", 512, 12 * 40, 2), ("A 512@24x80", 512, 24 * 80, 12), ("B 256@48x160", 256, 48 * 160, 12),
           ("C 128@96x320", 128, 96 * 320, 9), ("D 64@192x640", 64, 192 * 640, 7), ("E 32@384x1280", 32, 384 * 1280, 3)]


def bench(C, HW, B, has2, iters=30):
    dev = torch.device("cuda")
    per = B * HW * C * 2
    nset = max(2, min(8, int(600e6 // (per * (4 if not has2 else 6))) + 1))
    sets = []
    for _ in range(nset):
        y1 = K.new_act(B, C, HW, 1).normal_()
        y2 = K.new_act(B, C, HW, 1).normal_() if has2 else None
        dz = K.new_act(B, C, HW, 1).normal_()
        sets.append((y1, y2, dz))
    sc = (torch.rand(B, C, device=dev) >= 0.5).float() * 2 if has2 else None
    gm, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    res = {}
    split = int(os.environ.get("GN_BENCH_SPLIT", "1"))               # experiment: the backward in `split` batch groups, one after the other (reduce + apply of a group: its second read may stay in the Infinity Cache)
    gs = B // split
    for name in ("fwd", "bwd"):
        stats_keep = []
        for y1, y2, dz in sets:
            stats_keep.append([K._gn_forward(y1[g * gs:(g + 1) * gs], None if y2 is None else y2[g * gs:(g + 1) * gs], None if sc is None else sc[g * gs:(g + 1) * gs], gm, bt, 1e-5)[1] for g in range(split)])
        torch.cuda.synchronize()

        def body():
            for i in range(iters):
                y1, y2, dz = sets[i % nset]
                if name == "fwd":
                    K._gn_forward(y1, y2, sc, gm, bt, 1e-5)
                else:
                    for g in range(split):
                        K._gn_backward(dz[g * gs:(g + 1) * gs], y1[g * gs:(g + 1) * gs], None if y2 is None else y2[g * gs:(g + 1) * gs], None if sc is None else sc[g * gs:(g + 1) * gs],
                                       stats_keep[i % nset][g], gm, bt, 1e-5, has2, want_dbias=not has2)
        # replayed from a HIP graph: the Python enqueue (~25 us per call) must not be what is measured
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        K.begin_graph_capture()
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
        del graph
    return res


def main():
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        K.lib.mte_debug_set(int(k), int(v))
    K.set_compute_dtype("bf16")
    B = 8
    tot = 0.0
    for label, C, HW, calls in CLASSES:
        for has2 in (False, True):
            r = bench(C, HW, B, has2)
            el = B * HW * C
            nin = 2 if has2 else 1
            fb = el * 2 * (nin + 1)                  # minimum: read inputs once, write z
            bb = el * 2 * (nin + 1 + nin)            # minimum: read inputs + dz once, write d1 (+ d2)
            print("%-16s has2=%d  fwd %7.1f us (%5.0f GB/s of the 1-read minimum)   bwd %7.1f us (%5.0f GB/s)"
                  % (label, has2, r["fwd"], fb / r["fwd"] / 1e3, r["bwd"], bb / r["bwd"] / 1e3), flush=True)
            if not has2:
                tot += calls * (r["fwd"] + r["bwd"])
    print("weighted sum over the step's %d GroupNorm layers (has2=0 timings): %.2f ms" % (sum(c[3] for c in CLASSES), tot / 1e3))


if __name__ == "__main__":
    main()
