"""Builds libmte_hip.so (gfx950 only) in-tree with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libmte_hip.so")
SOURCES = ["conv_igemm.hip", "conv_patch.hip", "norm_act.hip", "pack3d.hip", "pack_fold.hip", "heads_misc.hip", "edge_loss.hip", "eval_metrics.hip", "dee_post.hip", "chamfer.hip", "canny.hip", "san.hip", "data_prep.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-ffp-contract=off"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every .hip for gfx950 and link the C-ABI shared library next to the sources."""
    hipcc = _hipcc()
    hdr = os.path.join(CSRC, "common.hpp")
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(CSRC, s.replace(".hip", ".o"))
        if force or _stale(obj, [src, hdr]):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
