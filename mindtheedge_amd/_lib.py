"""ctypes binding of libmte_hip.so.  The prototypes are read from include/mte_kernels.h, which is the
single source of truth for the C ABI.  There is NO fallback: if the library is missing or a kernel
returns an error code the caller gets an exception."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "mte_kernels.h")
LIB_PATH = os.environ.get("MTE_LIB_PATH") or os.path.join(_HERE, "csrc", "libmte_hip.so")   # (override: A/B of two builds)

_CTYPES = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double, "mte_stream_t": ctypes.c_void_p}
_ERRORS = {-1: "MTE_ERR_ARG (bad argument / unsupported shape)", -2: "MTE_ERR_LAUNCH (HIP launch failed)",
           -3: "MTE_ERR_UNSUPPORTED"}


class MteError(RuntimeError):
    pass


RETURNS = {}
# entry points that return a value (capability / size queries) instead of an error code
QUERIES = ("mte_conv2d_patch_supported", "mte_conv2d_patch_pack_elems", "mte_depth_metrics_workspace_bytes",
           "mte_chamfer_workspace_bytes", "mte_edge_loss_sums_elems", "mte_gn_fwd_is_single_pass",
           "mte_edge_loss_work_elems")


def parse_header(path=HEADER):
    """-> {name: [(ctype, argname), ...]} for every `int|long mte_*(...)` prototype (RETURNS[name] = restype)."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long)\s+(mte_\w+)\s*\(([^)]*)\)\s*;", text):
        RETURNS[m.group(2)] = ctypes.c_long if m.group(1) == "long" else ctypes.c_int
        args = []
        for a in m.group(3).split(","):
            a = " ".join(a.split())
            if "*" in a:
                args.append((ctypes.c_void_p, a.split("*")[-1].strip()))
            else:
                ty, name = a.rsplit(" ", 1)
                args.append((_CTYPES[ty.replace("const ", "").strip()], name))
        protos[m.group(2)] = args
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self._protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise MteError("libmte_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `python mindtheedge_amd/_build.py`; there is no CPU/eager fallback." % LIB_PATH)
            # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be in the process before this library so that
            # both resolve to ONE HIP runtime (loading /opt/rocm's copy first leaves torch's runtime without a device)
            import torch  # noqa: F401
            dll = ctypes.CDLL(LIB_PATH)
            for name, args in self._protos.items():
                fn = getattr(dll, name)          # AttributeError if the .so lacks a declared symbol
                fn.argtypes = [t for t, _ in args]
                fn.restype = RETURNS.get(name, ctypes.c_int)
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        if name.startswith("mte_"):
            fn = getattr(self.load(), name)
            if name in QUERIES:
                call = fn
            else:
                def call(*args):
                    rc = fn(*args)
                    if rc != 0:
                        raise MteError("%s failed: %s" % (name, _ERRORS.get(rc, rc)))
            self.__dict__[name] = call        # ~1000 launches per training step go through here: resolve each entry point once
            return call
        raise AttributeError(name)


lib = _Lib()
