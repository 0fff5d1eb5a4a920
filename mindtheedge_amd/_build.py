"""Builds libmte_hip.so (gfx950 only) in-tree with hipcc.  Cross-compiles without a GPU."""
import hashlib
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


def _digest(paths, extra=""):
    """Content hash of the inputs of one build product (sources + flags), so that staleness does not depend on
    file times: a fresh checkout / a pushed snapshot whose objects do not match its sources recompiles."""
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, digest):
    stamp = target + ".sha256"
    if not (os.path.exists(target) and os.path.exists(stamp)):
        return True
    with open(stamp) as f:
        return f.read().strip() != digest


def _stamp(target, digest):
    with open(target + ".sha256", "w") as f:
        f.write(digest + "\n")


def build(force=False, verbose=False):
    """Compile every .hip for gfx950 and link the C-ABI shared library next to the sources."""
    hipcc = _hipcc()
    hdr = os.path.join(CSRC, "common.hpp")
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(CSRC, s.replace(".hip", ".o"))
        dig = _digest([src, hdr], " ".join(FLAGS))
        if force or _stale(obj, dig):
            jobs.append(([hipcc] + FLAGS + ["-c", src, "-o", obj], obj, dig))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))

    def compile_one(job):
        cmd, obj, dig = job
        run(cmd)
        _stamp(obj, dig)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES]
    lib_dig = _digest(objs, "link")
    if force or jobs or _stale(LIB, lib_dig):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        _stamp(LIB, lib_dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
