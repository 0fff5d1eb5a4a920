"""Summarise rocprofv3 FETCH_SIZE / WRITE_SIZE counter_collection.csv files per kernel family.
usage: pmc_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <steps in each run> [json-out [source tag]]
With json-out: also writes the per-family figures bench.py quotes as `roofline.traffic` (profiles/pmc_traffic.json), stamped with the source tag and
the launch counts of the passes -- bench.py reports the figure only while its own run counts the same launches per step (round-4 advisor: a stale
numerator over a live denominator).  The conv family = igemm + patch + stem + generic / nine-tap weight gradients, as bench.py's timed set."""
import csv
import glob
import sys
from collections import defaultdict


def family(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    for key, fam in (("conv_igemm", "conv igemm (fwd+dgrad)"), ("splitk_finish", "conv igemm (fwd+dgrad)"), ("conv_wgrad", "conv wgrad generic"),
                     ("conv_patch_fwd", "conv patch fwd+dgrad"), ("conv_patch_wgrad", "conv patch wgrad"), ("conv_stem", "conv stem fwd+wgrad"),
                     ("pack_multi", "weight packs"), ("pack_weights", "weight packs"), ("unpack_wgrad", "wgrad unpack/reduce"), ("reduce_parts", "wgrad unpack/reduce"),
                     ("gn_", "GroupNorm+ELU"),
                     ("pack3d", "conv3d pack/unpack"), ("unpack3d", "conv3d pack/unpack"), ("invdepth", "invdepth head"), ("tap_wgrad", "invdepth head"),
                     ("adam", "adam"), ("fillBuffer", "memset"), ("copyBuffer", "memcpy")):
        if key in n:
            return fam
    return "other"


def load(d, counter, by_kernel=False):
    out = defaultdict(lambda: [0, 0.0])
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            e = out[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60] if by_kernel else family(r["Kernel_Name"])]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    return out


F, W, steps = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), float(sys.argv[3])
print("# per training step (T8: 8 x 384x1280), counters in KiB as rocprofv3 reports them; read = 2 x FETCH_SIZE (gfx950 correction)")
print("%-28s %10s %12s %12s %12s" % ("family", "launches", "read MB", "write MB", "total MB"))
tot_r = tot_w = 0.0
for fam in sorted(set(F) | set(W), key=lambda k: -(2 * F[k][1] + W[k][1])):
    r = 2.0 * F[fam][1] * 1024 / steps / 1e6
    w = W[fam][1] * 1024 / steps / 1e6
    tot_r += r
    tot_w += w
    print("%-28s %10.1f %12.1f %12.1f %12.1f" % (fam, max(F[fam][0], W[fam][0]) / steps, r, w, r + w))
print("%-28s %10s %12.1f %12.1f %12.1f" % ("all kernels", "", tot_r, tot_w, tot_r + tot_w))

print()
print("# the same per kernel (top 40 by total bytes)")
Fk, Wk = load(sys.argv[1], "FETCH_SIZE", True), load(sys.argv[2], "WRITE_SIZE", True)
for k in sorted(set(Fk) | set(Wk), key=lambda k: -(2 * Fk[k][1] + Wk[k][1]))[:40]:
    r = 2.0 * Fk[k][1] * 1024 / steps / 1e6
    w = Wk[k][1] * 1024 / steps / 1e6
    print("%-62s %8.1f %10.1f %10.1f" % (k, max(Fk[k][0], Wk[k][0]) / steps, r, w))

if len(sys.argv) > 4:
    import json
    conv = ("conv igemm (fwd+dgrad)", "conv wgrad generic", "conv patch fwd+dgrad", "conv patch wgrad", "conv stem fwd+wgrad")
    mb = lambda fams: sum((2.0 * F[f][1] + W[f][1]) * 1024 / steps / 1e6 for f in fams)
    out = {"mode": "train", "batch": 8, "height": 384, "width": 1280, "dtype": "bf16",
           "source": sys.argv[5] if len(sys.argv) > 5 else "",
           "conv_family_MB_per_step": round(mb(conv), 1),
           "conv_family_kernel_launches_per_step": sum(max(F[f][0], W[f][0]) for f in conv) / steps,
           "gn_family_MB_per_step": round(mb(("GroupNorm+ELU",)), 1),
           "gn_family_kernel_launches_per_step": max(F["GroupNorm+ELU"][0], W["GroupNorm+ELU"][0]) / steps,
           "all_kernels_MB_per_step": round(tot_r + tot_w, 1)}
    with open(sys.argv[4], "w") as f:
        json.dump(out, f, indent=1)
