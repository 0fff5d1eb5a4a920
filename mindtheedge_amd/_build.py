"""Builds the gfx950 kernel libraries in-tree with hipcc (cross-compiles without a GPU):

  csrc/libmte_hip.so      the product: exports exactly the integration surface of include/mte_kernels.h
  csrc/libmte_hip_dev.so  the same sources with -DMTE_DEV: adds mte_debug_set (launch-geometry / kernel-variant knobs) and the
                          main-loop ablation arms of the implicit GEMM, for tools/ and the kernel-variant cross-checks in tests/
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libmte_hip.so")
SOURCES = ["conv_igemm.hip", "conv_igemm8.hip", "conv_wgrad9.hip", "tap_wgrad.hip", "conv_patch.hip", "conv_stem.hip", "norm_act.hip", "pack3d.hip", "pack_fold.hip", "heads_misc.hip", "edge_loss.hip", "eval_metrics.hip", "dee_post.hip", "chamfer.hip", "canny.hip", "san.hip", "data_prep.hip"]
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


DEV_LIB = os.path.join(CSRC, "libmte_hip_dev.so")


def build(force=False, verbose=False, dev=True):
    """Compile every .hip for gfx950 and link the C-ABI shared library next to the sources (and, dev=True, its -DMTE_DEV twin)."""
    hipcc = _hipcc()
    hdr = os.path.join(CSRC, "common.hpp")
    hdr2 = os.path.join(CSRC, "conv_args.hpp")
    os.makedirs(os.path.join(CSRC, "dev"), exist_ok=True)
    variants = [("", [], LIB)] + ([("dev", ["-DMTE_DEV"], DEV_LIB)] if dev else [])
    jobs = []
    for sub, extra, _ in variants:
        for s in SOURCES:
            src, obj = os.path.join(CSRC, s), os.path.join(CSRC, sub, s.replace(".hip", ".o"))
            dig = _digest([src, hdr, hdr2], " ".join(FLAGS + extra))
            if force or _stale(obj, dig):
                jobs.append(([hipcc] + FLAGS + extra + ["-c", src, "-o", obj], obj, dig))

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
        with ThreadPoolExecutor(max_workers=min(int(os.environ.get("MTE_BUILD_JOBS", "6")), len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    for sub, _, target in variants:
        objs = [os.path.join(CSRC, sub, s.replace(".hip", ".o")) for s in SOURCES]
        lib_dig = _digest(objs, "link")
        if force or _stale(target, lib_dig):
            run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs)
            _stamp(target, lib_dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
