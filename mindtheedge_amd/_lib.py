"""ctypes binding of libmte_hip.so.  The prototypes are read from include/mte_kernels.h, which is the
single source of truth for the C ABI.  There is NO fallback: if the library is missing or a kernel
returns an error code the caller gets an exception."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "mte_kernels.h")
DEV_LIB_PATH = os.path.join(_HERE, "csrc", "libmte_hip_dev.so")      # -DMTE_DEV build: + mte_debug_set (tools/, kernel-variant tests)
# MTE_LIB_PATH: A/B of two builds; MTE_USE_DEV_LIB=1: tools that turn development knobs run on the dev build from the start
_USE_DEV = bool(os.environ.get("MTE_USE_DEV_LIB"))
LIB_PATH = os.environ.get("MTE_LIB_PATH") or (DEV_LIB_PATH if _USE_DEV else os.path.join(_HERE, "csrc", "libmte_hip.so"))

_CTYPES = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double, "mte_stream_t": ctypes.c_void_p}
_ERRORS = {-1: "MTE_ERR_ARG (bad argument / unsupported shape)", -2: "MTE_ERR_LAUNCH (HIP launch failed)",
           -3: "MTE_ERR_UNSUPPORTED"}


class MteError(RuntimeError):
    pass


RETURNS = {}
# entry points that return a value (capability / size queries) instead of an error code
QUERIES = ("mte_conv2d_patch_supported", "mte_conv2d_stem_supported", "mte_conv2d_patch_wgrad_supported", "mte_conv2d_wgrad_nine_tap", "mte_conv2d_patch_fwd_gn_elems", "mte_conv2d_patch_pack_elems", "mte_depth_metrics_workspace_bytes",
           "mte_chamfer_workspace_bytes", "mte_edge_loss_sums_elems", "mte_gn_fwd_is_single_pass", "mte_gn_fwd_is_single_pass_b",
           "mte_edge_loss_work_elems", "mte_rank1_conv_bwd_records_elems", "mte_conv2d_patch_fwd_rank1_ok", "mte_gn_stats_elems", "mte_device_error_poll", "mte_invdepth_bwd_weight_workspace_elems", "mte_sparse_site_list_workspace_elems")


def parse_header(path=HEADER, dev=False):
    """-> {name: [(ctype, argname), ...]} for every `int|long mte_*(...)` prototype (RETURNS[name] = restype).
    dev=False: the integration surface (what libmte_hip.so exports); dev=True: + the `#ifdef MTE_DEV` section."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    if not dev:
        text = re.sub(r"#ifdef MTE_DEV.*?#endif", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long)\s+(mte_\w+)\s*\(([^)]*)\)\s*;", text):
        RETURNS[m.group(2)] = ctypes.c_long if m.group(1) == "long" else ctypes.c_int
        args = []
        for a in ([] if m.group(3).strip() in ("", "void") else m.group(3).split(",")):
            a = " ".join(a.split())
            if "*" in a:
                args.append((ctypes.c_void_p, a.split("*")[-1].strip()))
            else:
                ty, name = a.rsplit(" ", 1)
                args.append((_CTYPES[ty.replace("const ", "").strip()], name))
        protos[m.group(2)] = args
    return protos


class _Lib:
    def __init__(self, path=None, dev=None):
        self._dll = None
        self._path = path or LIB_PATH
        self._dev = (self._path == DEV_LIB_PATH) if dev is None else bool(dev)
        self._protos = parse_header(dev=self._dev)
        self._options = {}               # mte_set_option calls, re-applied when another build is switched in
        self._loaded = {}                # path -> CDLL

    def load(self):
        if self._dll is None:
            if not os.path.exists(self._path):
                raise MteError("%s is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `python mindtheedge_amd/_build.py`; there is no CPU/eager fallback." % (os.path.basename(self._path), self._path))
            # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be in the process before this library so that
            # both resolve to ONE HIP runtime (loading /opt/rocm's copy first leaves torch's runtime without a device)
            import torch  # noqa: F401
            dll = self._loaded.get(self._path)
            if dll is None:
                dll = self._loaded[self._path] = ctypes.CDLL(self._path)
            for name, args in self._protos.items():
                fn = getattr(dll, name)          # AttributeError if the .so lacks a declared symbol
                fn.argtypes = [t for t, _ in args]
                fn.restype = RETURNS.get(name, ctypes.c_int)
            self._dll = dll
            for opt, val in self._options.items():
                dll.mte_set_option(opt, val)
        return self._dll

    def switch(self, path, dev):
        """Route every later call to another build of the library (development: see dev_library)."""
        for k in [k for k in self.__dict__ if k.startswith("mte_")]:
            del self.__dict__[k]                 # cached entry points of the previous build
        self._dll, self._path, self._dev = None, path, bool(dev)
        self._protos = parse_header(dev=self._dev)

    def set_option(self, option, value, lazy=False):
        """lazy: only remember the option when the library is not in the process yet (applied by load())"""
        self._options[int(option)] = int(value)
        if not lazy or self._dll is not None:
            self.mte_set_option(int(option), int(value))

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


class dev_library:
    """``with dev_library(): lib.mte_debug_set(key, value); ...`` -- run on libmte_hip_dev.so, the -DMTE_DEV build of the same
    sources (kernel-variant cross-checks in tests/, A/B tools).  The shipped library does not export mte_debug_set."""

    def __enter__(self):
        self._prev = (lib._path, lib._dev)
        lib.switch(DEV_LIB_PATH, True)
        return lib

    def __exit__(self, *exc):
        lib.switch(*self._prev)
        return False
