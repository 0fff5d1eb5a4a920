"""torch.autograd.Function wrappers over the C ABI (include/mte_kernels.h).

Activations travel between layers as torch tensors of logical shape [B, C, H, W] whose memory is NHWC
(``channels_last`` strides, possibly a channel slice of a wider buffer) in the compute dtype (bf16, or
fp32 in validation mode).  PyTorch is used for device memory, streams and the autograd tape only: every
arithmetic step of the hot path is a hand-written gfx950 kernel in libmte_hip.so.  There is no fallback
path -- a missing library or an unsupported shape raises.
"""
import ctypes
import itertools
import math
import os
import weakref

import torch

from ._lib import lib, MteError

DT_BF16, DT_F32 = 0, 1
_state = {"compute_dtype": torch.bfloat16, "weights_epoch": 0}


def set_compute_dtype(dtype):
    """'bf16' (default; MFMA bf16, fp32 accumulate) or 'fp32' (exact fp32 MFMA; validation mode)."""
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "f32": torch.float32}.get(dtype, dtype)
    assert dtype in (torch.bfloat16, torch.float32)
    _state["compute_dtype"] = dtype


def compute_dtype():
    return _state["compute_dtype"]


def bump_weights_epoch():
    """Invalidate every cached bf16/fp32 weight pack (call after an optimizer step that writes parameters
    behind autograd's version counters, e.g. the fused flat Adam)."""
    _state["weights_epoch"] += 1


def weights_epoch():
    return _state["weights_epoch"]


class GradSink:
    """Lets backward kernels write parameter gradients straight into pre-allocated storage (the flat gradient buffer
    of trainers.data_parallel.FlatParameters) instead of returning fresh tensors that autograd then adds into .grad:
    on the benchmarked path every parameter receives exactly one gradient per backward, so a plain store into the zeroed
    buffer IS the accumulation.  ``ready(p)`` replaces the post-accumulate-grad hook (bucketed all-reduce trigger).

    A store is only an accumulation ONCE per zeroed buffer.  The sink therefore hands a parameter's view out for one store
    per ``reset()`` (= optimizer.zero_grad()): a second gradient for the same tensor -- a second backward pass without
    zero_grad, or a parameter used twice in one graph -- goes back to autograd, which ADDS it into ``.grad`` (the same flat
    memory) on the main stream.  The first store may have been queued on the weight-gradient side stream, so before handing the
    second gradient to autograd the sink makes the current stream wait for the side streams (``join_side_stream``): the add can
    never overtake the store (round-3 advisor finding).  What the sink canNOT repair is the ready() announcement: it went out with
    the first, incomplete gradient, so with a consumer attached (``on_ready``: the bucketed all-reduce) a second gradient for an
    already announced parameter raises instead of reducing half a gradient.  A forward pass that knows it shares parameters
    between two sub-graphs (PackNetSAN01's RGB and RGB+LiDAR passes, reference networks/depth/PackNetSAN01.py:324-338) calls
    ``suspend()`` BEFORE its backward: until the next ``reset()`` EVERY gradient takes the autograd route, so each parameter's
    AccumulateGrad node runs once with the sum of both uses and the all-reduce trigger (post-accumulate hook) never fires on half
    a gradient."""

    def __init__(self):
        self.views = {}          # param.data_ptr() -> flat fp32 gradient view
        self.on_ready = None
        self.written = set()     # parameters whose view already holds this step's (first) gradient
        self.announced = set()   # ... and whose gradient was announced through ready()
        self.suspended = False
        self._suspend_resets = 0

    def register(self, p, view):
        self.views[p.data_ptr()] = view

    def lookup(self, p):
        """the view a gradient of `p` may be STORED into, or None (not registered / already written / suspended)"""
        if self.suspended:
            return None
        key = p.data_ptr()
        v = self.views.get(key)
        if v is not None and key in self.written:
            # a second gradient for a parameter whose view already took this step's store: autograd will add it into the same memory
            if key in self.announced and self.on_ready is not None:
                raise MteError("a parameter received a second gradient after its first one was announced to the gradient all-reduce: "
                               "call kernels.suspend_grad_sink() in the forward pass of a graph that uses parameters twice")
            join_side_stream()
            return None
        if v is not None and p.grad is not None and p.grad.data_ptr() == v.data_ptr() and v.shape == p.shape:
            return v
        return None

    def claim(self, p):
        """lookup + mark as written: the caller stores this step's gradient of `p` into the returned view"""
        v = self.lookup(p)
        if v is not None:
            self.written.add(p.data_ptr())
        return v

    def reset(self):
        """the gradient buffer was cleared (zero_grad): every view may take one store again.  A suspension survives the FIRST
        zero_grad after it (forward -> zero_grad -> backward is a legal order: the backward that needs the suspension is still to
        come); the optimizer step (end_step) or a second zero_grad ends it -- staying suspended one step too long costs speed only."""
        self.written.clear()
        self.announced.clear()
        if self.suspended:
            self._suspend_resets += 1
            if self._suspend_resets >= 2:
                self.suspended = False

    def end_step(self):
        """the optimizer consumed the gradients: a forward pass's suspension ends here"""
        self.suspended = False

    def suspend(self):
        self.suspended = True
        self._suspend_resets = 0

    def ready(self, p):
        self.announced.add(p.data_ptr())
        if self.on_ready is not None:
            self.on_ready(p)


_sink = {"active": None}


def set_grad_sink(sink):
    _sink["active"] = sink


def suspend_grad_sink():
    """called by a forward pass whose graph uses parameters more than once (see GradSink)"""
    if _sink["active"] is not None:
        _sink["active"].suspend()


def _grad_dst(p, zero=False):
    """(tensor to write the gradient of `p` into, True if it is the sink's storage).  zero: the kernel ACCUMULATES into
    the destination (the sink's flat gradient buffer is cleared by zero_grad; a private one comes zeroed)."""
    sk = _sink["active"]
    if sk is not None:
        v = sk.claim(p)
        if v is not None:
            return v, True
    if zero:
        return _zeros(p.shape, torch.float32, p.device), False
    return torch.empty(p.shape, dtype=torch.float32, device=p.device), False


def _grad_ret(p, t, sunk):
    if sunk:
        _sink["active"].ready(p)
        return None
    return t


def _dt(t):
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float32:
        return DT_F32
    raise MteError("unsupported activation dtype %s" % t.dtype)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """raw handle of the current HIP stream.  ~1000 calls per training step: torch.cuda.current_stream() builds a Stream object
    through several Python layers (2.5 ms of host time per forward pass); the C accessors return the same handle directly."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(t):
    if not t.is_cuda:
        raise MteError("mindtheedge_amd kernels run on an MI355X only; got a %s tensor (no CPU fallback)" % t.device)


def round8(c):
    return (c + 7) // 8 * 8


def new_act(B, C, H, W, dtype=None, device="cuda"):
    """Uninitialised NHWC activation with logical shape [B,C,H,W]."""
    return torch.empty_strided((B, C, H, W), (H * W * C, 1, W * C, C), dtype=dtype or compute_dtype(), device=device)   # (one operator, not empty + permute)


def alias_of(t):
    """A fresh tensor object over the same memory (not an autograd view of `t`): lets an op write its result into a
    caller-chosen place (a channel slice of a decoder concat buffer) and still return a tensor it owns."""
    return torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage(), t.storage_offset(), t.size(), t.stride())


def channel_slice(buf, c0, c1):
    """Channel block [c0, c1) of an NHWC activation as an independent tensor object (see alias_of)."""
    return alias_of(buf[:, c0:c1])


def is_act(t):
    if t.dim() != 4 or t.dtype not in (torch.bfloat16, torch.float32):
        return False
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if C > 1 and sc != 1:
        return False
    ld = sw if W > 1 else (sh if H > 1 else (sb if B > 1 else C))
    ok = ld >= C and (H == 1 or W == 1 or sh == W * ld) and (B == 1 or sb == H * W * ld)
    per16 = 8 if t.dtype == torch.bfloat16 else 4
    return bool(ok and ld % per16 == 0 and t.data_ptr() % 16 == 0)


def as_act(t, dtype=None):
    """Returns t as an NHWC activation in `dtype` (layout/dtype conversion only when necessary)."""
    dtype = dtype or t.dtype
    if t.dtype == dtype and is_act(t):
        return t
    out = new_act(t.shape[0], t.shape[1], t.shape[2], t.shape[3], dtype, t.device)
    out.copy_(t)
    return out


def _pl(t):
    """(device pointer, elements per pixel) of an NHWC activation."""
    if t.dim() == 4:                    # fast path (~2000 calls per training step): the plain [B,C,H,W]-over-NHWC case
        B, C, H, W = t.shape
        sb, sc, sh, sw = t.stride()
        if sc == 1 and W > 1 and H > 1 and sh == W * sw and (B == 1 or sb == H * sh) and sw >= C:
            dt = t.dtype
            if (dt is torch.bfloat16 and not (sw & 7)) or (dt is torch.float32 and not (sw & 3)):
                p = t.data_ptr()
                if not (p & 15):
                    return p, sw
    if not is_act(t):
        raise MteError("expected an NHWC activation, got shape %s strides %s" % (tuple(t.shape), t.stride()))
    B, C, H, W = t.shape
    ld = t.stride(3) if W > 1 else (t.stride(2) if H > 1 else (t.stride(0) if B > 1 else C))
    return t.data_ptr(), ld


def _ptr(t):
    return 0 if t is None else t.data_ptr()


_ELEM_SIZE = {torch.float64: 8, torch.float32: 4, torch.bfloat16: 2, torch.float16: 2, torch.uint8: 1, torch.int32: 4, torch.int64: 8}


class _ZeroArena:
    """Small zero-initialised device buffers (GroupNorm statistics / reductions / bias-gradient vectors) carved from
    8 MiB chunks that are cleared by ONE fill each: replaces ~140 per-layer 2-32 KiB memset launches per training step
    (the library runs with MTE_OPT_GN_PREZEROED).  A chunk lives as long as any buffer carved from it."""
    CHUNK = 8 << 20

    def __init__(self):
        self.chunks = {}                 # (device, stream) -> [uint8 chunk, offset, capture id or None]
        self.captured = []               # chunks created under HIP-graph capture (kept alive for the graphs that fill them)
        self.enabled = False

    def zeros(self, shape, dtype, device):
        if not self.enabled:             # from now on the library trusts GroupNorm accumulation buffers to arrive zeroed
            lib.set_option(0, 1)
            lib.set_option(1, 1)         # ... and the loss workspaces (also carved from here)
            if os.environ.get("MTE_HANDOFF_FENCES"):
                lib.set_option(2, 1 if os.environ["MTE_HANDOFF_FENCES"] == "1" else 0)
            self.enabled = True
            _device_errors_init()
        n = 1
        for d in shape:
            n *= d
        nbytes = (n * _ELEM_SIZE[dtype] + 255) & ~255
        if nbytes > self.CHUNK // 4:
            return torch.zeros(shape, dtype=dtype, device=device)
        capturing = torch.cuda.is_current_stream_capturing()
        key = (device if isinstance(device, str) else (device.type, device.index), _stream())
        c = self.chunks.get(key)
        # Under HIP-graph capture a chunk must have been created INSIDE the running capture: only then is its one fill a node
        # of this graph and the buffers are zero again at every replay.  A chunk left over from eager work or from an
        # earlier capture (same capture stream, other graph) has no fill node here: buffers carved from it would accumulate
        # on stale statistics from the second replay on.
        if c is None or c[1] + nbytes > self.CHUNK or (capturing and c[2] != self.capture_id) or (not capturing and c[2] is not None):
            if c is not None and c[2] is not None:
                self.captured.append(c[0])                   # a captured graph keeps replaying into this chunk: never hand it back
            c = self.chunks[key] = [torch.zeros((self.CHUNK,), dtype=torch.uint8, device=device), 0,
                                    self.capture_id if capturing else None, {}]
        typed = c[3].get(dtype)
        if typed is None:
            typed = c[3][dtype] = c[0].view(dtype)
        strides, acc = [], 1
        for d in reversed(shape):
            strides.append(acc)
            acc *= d
        out = typed.as_strided(shape, strides[::-1], c[1] // _ELEM_SIZE[dtype])     # (one operator per buffer: ~190 of them per training step)
        c[1] += nbytes
        return out

    capture_id = 0

    def begin_capture(self):
        """call right before a HIP-graph capture starts: chunks of earlier captures are never carved again"""
        self.capture_id += 1


_arena = _ZeroArena()
_DEVERR_NAMES = {1: "GroupNorm cluster kernel (forward): a workgroup of a cluster never arrived",
                 2: "GroupNorm cluster kernel (backward): a workgroup of a cluster never arrived"}


def _device_errors_init():
    """allocate the device error word of the library build in use (include/mte_kernels.h: mte_device_error_init; a no-op from the second call on) --
    never inside a stream capture.  -> True when the word exists"""
    if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        lib.mte_device_error_init()
        return True
    return False


def check_device_errors():
    """Raise if a kernel of this process reported through the device error word since the last call (a bounded inter-workgroup wait that gave up:
    its results are wrong by construction).  Costs one read of host memory, no synchronisation: the trainer and the optimizer call it once per
    step, so an error surfaces at the end of the step that caused it or of the next one -- FusedAdam.step() polls AFTER it has queued the update and the
    weight re-pack, so by then the parameters carry the failed step: the caller restores its last checkpoint.  The inference entry points poll too
    (infer_edges.infer_depth after waiting for its stream, ModelWrapper.depth without waiting)."""
    code = int(lib.mte_device_error_poll()) if _device_errors_init() else 0
    if code:
        what = "; ".join(v for k, v in _DEVERR_NAMES.items() if code & k) or "code %d" % code
        raise MteError("device error word set (0x%x): %s -- the results of that step are invalid; when this is raised by the optimizer, the parameter "
                       "update of the failed step has already been applied: restore the last checkpoint" % (code, what))


def begin_graph_capture():
    """Library-side state a HIP-graph capture must not inherit (see _ZeroArena.zeros): zero-arena chunks and the pooled
    Dropout2d draws (a pool left over from eager steps would be baked into the graph as constants; a pool drawn INSIDE the
    capture is a node whose Philox offset torch advances at every replay)."""
    _arena.begin_capture()
    _dropout_pool.clear()


def end_graph_capture():
    """After a capture: drop the references eager code would otherwise keep into the graph's private memory pool."""
    _dropout_pool.clear()


def _zeros(shape, dtype, device):
    return _arena.zeros(tuple(shape), dtype, device)


# --------------------------------------------------------------------------------------------------
# weight packs
# --------------------------------------------------------------------------------------------------
class WeightPack:
    """Kernel-ready copies of one OIHW fp32 conv weight: forward [Cout][taps][Cin_p] and data-gradient
    [Cin_p][taps rot180][Cout] packs in the compute dtype.  Re-packed when the parameter changes."""

    _live = []                          # weak references to every pack (prefetch_weight_packs walks them)
    _seq = itertools.count()

    def __init__(self):
        self.key = None
        self.wf = self.wb = None
        self.pf = self.pb = None        # fragment-block packs for the LDS-patch kernels (bf16, C_out <= 64)
        self.pkey = None
        self._last = None               # (weakref to the parameter, dtype) of the latest request
        self._order = -1                # when it was last requested (forward execution order)
        self._event = None              # set by prefetch_weight_packs: side-stream event that makes the packs valid
        WeightPack._live.append(weakref.ref(self))

    def _sync_prefetch(self):
        if self._event is not None:
            cur = torch.cuda.current_stream()
            waited = _side.setdefault("last_waited", {})      # packs are prefetched in groups that share one event: one wait per stream
            if waited.get(cur.cuda_stream) is not self._event:
                cur.wait_event(self._event)
                waited[cur.cuda_stream] = self._event
            self._event = None

    def _patch_pack(self, w, which):
        wf, wb = self.wf, self.wb
        cout, cin, kh, kw = w.shape
        cin_p = round8(cin)
        if which == 'f':
            n = lib.mte_conv2d_patch_pack_elems(cin_p, cout, kh, kw)
            if self.pf is None or self.pf.numel() != n:
                self.pf = torch.empty((n,), dtype=torch.bfloat16, device=w.device)
            lib.mte_conv2d_patch_repack(wf.data_ptr(), self.pf.data_ptr(), cin_p, cout, kh, kw, _stream())
        else:
            n = lib.mte_conv2d_patch_pack_elems(cout, cin_p, kh, kw)
            if self.pb is None or self.pb.numel() != n:
                self.pb = torch.empty((n,), dtype=torch.bfloat16, device=w.device)
            lib.mte_conv2d_patch_repack(wb.data_ptr(), self.pb.data_ptr(), cout, cin_p, kh, kw, _stream())

    def get_patch(self, w, which):
        """which = 'f' (forward: N = Cout, K over Cin) or 'b' (data gradient: N = Cin_p, K over Cout)."""
        self.get(w, torch.bfloat16, which == 'b')
        if self.pkey != self.key:
            self.pf_ok = self.pb_ok = False
            self.pkey = self.key
        if which == 'f':
            if not self.pf_ok:
                self._patch_pack(w, 'f')
                self.pf_ok = True
            return self.pf
        if not self.pb_ok:
            self._patch_pack(w, 'b')
            self.pb_ok = True
        return self.pb

    pf_ok = pb_ok = False
    _want_pf = _want_pb = False

    def _pack(self, w, dtype, need_bwd):
        """(re)build the forward (and data-gradient) pack on the current stream, re-using the buffers when they fit"""
        cout, cin, kh, kw = w.shape
        cin_p = round8(cin)
        dtc = DT_BF16 if dtype == torch.bfloat16 else DT_F32
        if self.wf is None or self.wf.dtype != dtype or tuple(self.wf.shape) != (cout, kh * kw, cin_p):
            self.wf = torch.empty((cout, kh * kw, cin_p), dtype=dtype, device=w.device)
            self.wb = None
        if need_bwd and (self.wb is None or self.wb.dtype != dtype):
            self.wb = torch.empty((cin_p, kh * kw, cout), dtype=dtype, device=w.device)
        lib.mte_pack_conv_weights(w.detach().contiguous().data_ptr(), self.wf.data_ptr(), _ptr(self.wb) if need_bwd else 0,
                                  cout, cin, kh, kw, cin_p, cout, dtc, _stream())
        self.has_bwd = bool(need_bwd)

    has_bwd = False

    def get(self, w, dtype, need_bwd):
        key = (w.data_ptr(), w._version, weights_epoch(), dtype, tuple(w.shape))
        self._last = (weakref.ref(w), dtype)
        self._order = next(WeightPack._seq)
        if key != self.key:
            if self._event is not None:                      # a prefetch launch may still be writing these buffers on the side stream (advisor, round 4)
                torch.cuda.current_stream().wait_event(self._event)
            self._event = None
            self._pack(w, dtype, need_bwd)
            self.key = key
        else:
            self._sync_prefetch()
            if need_bwd and not self.has_bwd:
                cout, cin, kh, kw = w.shape
                cin_p = round8(cin)
                if self.wb is None:
                    self.wb = torch.empty((cin_p, kh * kw, cout), dtype=dtype, device=w.device)
                lib.mte_pack_conv_weights_bwd(self.wf.data_ptr(), self.wb.data_ptr(), cout, kh, kw, cin_p,
                                              DT_BF16 if dtype == torch.bfloat16 else DT_F32, _stream())
                self.has_bwd = True
        return self.wf, (self.wb if self.has_bwd else None)


class _PackJob(ctypes.Structure):           # mte_pack_job of include/mte_kernels.h
    _fields_ = [("w", ctypes.c_void_p), ("wf", ctypes.c_void_p), ("wb", ctypes.c_void_p), ("pf", ctypes.c_void_p), ("pb", ctypes.c_void_p),
                ("Cout", ctypes.c_int), ("Cin", ctypes.c_int), ("taps", ctypes.c_int), ("Cin_p", ctypes.c_int),
                ("end_f", ctypes.c_int), ("end_b", ctypes.c_int), ("end_pf", ctypes.c_int), ("end_pb", ctypes.c_int)]


_PACK_CHUNKS = (16, 48)          # jobs per launch: a short first launch (stem .. conv2: the layers the next forward pass needs at once), then 48s
_pack_jobs = {"sig": None, "chunks": []}


def _pack_job_chunks(packs):
    """ctypes job arrays for mte_pack_conv_weights_multi, cached while no pack changed its buffers: [(array, n, dtype code, packs)]"""
    sig = (len(packs),) + tuple((id(pk), pk._last[0]().data_ptr(), pk.wf.data_ptr(), _ptr(pk.wb) if pk.has_bwd else 0,
                                 _ptr(pk.pf) if pk._want_pf else 0, _ptr(pk.pb) if pk._want_pb else 0) for pk in packs)
    if _pack_jobs["sig"] == sig:
        return _pack_jobs["chunks"]
    chunks = []
    for dt in (torch.bfloat16, torch.float32):               # one element type per launch (in practice all packs share one)
        part = [pk for pk in packs if pk._last[1] == dt and pk.wf is not None and pk.wf.dtype == dt]
        i = 0
        while i < len(part):
            group = part[i:i + (_PACK_CHUNKS[0] if i == 0 else _PACK_CHUNKS[1])]
            i += len(group)
            arr = (_PackJob * len(group))()
            for o, pk in zip(arr, group):
                w = pk._last[0]()
                cout, cin, kh, kw = w.shape
                o.w, o.wf = w.data_ptr(), pk.wf.data_ptr()
                o.wb = pk.wb.data_ptr() if pk.has_bwd else None
                o.pf = pk.pf.data_ptr() if pk._want_pf else None
                o.pb = pk.pb.data_ptr() if pk._want_pb else None
                o.Cout, o.Cin, o.taps, o.Cin_p = cout, cin, kh * kw, round8(cin)
            chunks.append((arr, len(group), DT_BF16 if dt == torch.bfloat16 else DT_F32, group))
    _pack_jobs["sig"], _pack_jobs["chunks"] = sig, chunks
    return chunks


def prefetch_weight_packs():
    """Re-pack every conv weight that the last forward/backward used, now, on the side stream (called by the optimizer
    right after it changed the parameters): the packs of a step then overlap the first layers of the next forward instead of
    sitting in front of each layer's convolution.  All of them -- forward, data-gradient and LDS-patch fragment packs of ~100
    layers -- are rebuilt in place by three launches of ONE multi-tensor kernel (mte_pack_conv_weights_multi; the per-layer launches
    they replace were ~400 per step, launch-bound: 0.9 ms of GPU time and ~2 ms of host time).  Each launch's packs share an event
    that their next user waits on.  With the side stream switched off (serial profiling runs) the same launches go to the current stream."""
    packs = []
    for r in WeightPack._live:
        pk = r()
        if pk is not None and pk._last is not None and pk.key is not None and pk._last[0]() is not None and pk._last[0]().is_cuda:
            packs.append(pk)
    WeightPack._live = [r for r in WeightPack._live if r() is not None]
    if not packs:
        return
    packs.sort(key=lambda pk: pk._order)
    if _side["enabled"]:
        if not _side["streams"]:
            _side["streams"] = [_new_side_stream() for _ in range(max(1, _SIDE_STREAMS))]
        side = _side["streams"][0]
        ev = torch.cuda.Event()
        ev.record()                                          # the parameter update (and every earlier user of the packs)
        side.wait_event(ev)
    else:
        side = torch.cuda.current_stream()
    with torch.cuda.stream(side):
        _refold_all()                                        # folded pack weights first: their packs are rebuilt below
        _refresh_main_weights(packs)                         # ... and the main-channel copies of the rank-1 iconv layers
        for pk in packs:
            pk._want_pf = pk.pf_ok and pk.pkey == pk.key and pk.pf is not None
            pk._want_pb = pk.pb_ok and pk.pkey == pk.key and pk.pb is not None and pk.has_bwd
        st = _stream()
        odd = [pk for pk in packs if not (pk._last[0]().is_contiguous() and pk._last[0]().dtype == torch.float32)]
        if odd:                                              # (not the flat fp32 master layout: per-layer path)
            packs = [pk for pk in packs if pk not in odd]
            for pk in odd:
                w, dtype = pk._last[0](), pk._last[1]
                pk._pack(w, dtype, pk.has_bwd)
                pk.key = (w.data_ptr(), w._version, weights_epoch(), dtype, tuple(w.shape))
                pk.pkey, pk.pf_ok, pk.pb_ok, pk._event = None, False, False, None
        for arr, n, dtc, group in _pack_job_chunks(packs):
            lib.mte_pack_conv_weights_multi(ctypes.addressof(arr), n, dtc, st)
            ev = None
            if _side["enabled"]:
                ev = torch.cuda.Event()
                ev.record(side)
            for pk in group:
                w, dtype = pk._last[0](), pk._last[1]
                pk.key = (w.data_ptr(), w._version, weights_epoch(), dtype, tuple(w.shape))
                pk.pkey = pk.key
                pk.pf_ok, pk.pb_ok = pk._want_pf, pk._want_pb
                pk._event = ev
    if _side["enabled"] and torch.cuda.is_current_stream_capturing():
        # a HIP-graph capture must end with every forked stream joined; inside a graph the packs are DAG nodes that depend on
        # the optimizer only, so the replay still overlaps them with whatever else is ready
        torch.cuda.current_stream().wait_stream(side)
        for pk in packs:
            pk._event = None


SPLITK_SLABS = 8
_splitk_ws = {}


def _splitk_workspace(M, N, device):
    """fp32 scratch for split-K: one [M][N] slab per split (each split stores its partial tile, a finish kernel adds the slabs in
    order -- bit-reproducible, nothing to clear), offered only where the output has few 128x128 tiles (the library decides)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= 384:
        return None, 0
    n = SPLITK_SLABS * M * N
    if torch.cuda.is_current_stream_capturing():             # a graph keeps its own allocation alive (private pool)
        return torch.empty((n,), dtype=torch.float32, device=device), n
    # one scratch buffer per stream, grown on demand: launches of a stream run in order, so the next conv's slabs cannot overtake the
    # previous one's finish kernel (round-3 advisor: a fresh up-to-200 MB tensor per call was allocator churn for nothing)
    key = (str(device), _stream())
    ws = _splitk_ws.get(key)
    if ws is None or ws.numel() < n:
        ws = _splitk_ws[key] = torch.empty((n,), dtype=torch.float32, device=device)
    return ws, n


_cfg = {"patch_kernels": True, "pack_folding": True, "pack_fold_max_overhead": 0.30,
        "no_fork_accumulate": bool(os.environ.get("MTE_NO_FORK_ACCUM")),
        "patch_wgrad_slabs": not os.environ.get("MTE_PATCH_WGRAD_ATOMICS"),
        # decoder iconv3 / iconv2 / iconv1: the up-sampled inverse-depth input channel as a rank-1 term beside a GEMM over the other 192 / 96 / 64
        # channels (ConvGnEluInvFn) instead of a 65th channel that costs the GEMM kernels a whole 32-channel slice
        "split_inv_channel": os.environ.get("MTE_SPLIT_INV", "1") == "1",
        # ... and that term formed inside the LDS-patch forward's store loop where the layer runs on it (iconv1, iconv2): MTE_FUSE_INV=0 = written first, accumulated onto
        "fuse_inv_term": os.environ.get("MTE_FUSE_INV", "1") == "1",
        # residual blocks: the 1x1 shortcut's data gradient rides the 3x3 conv1's data-gradient launch as extra K-steps (mte_conv2d_patch_fwd_plus1x1) where
        # that launch runs on the LDS-patch kernel -- no stand-alone 1x1 launch, no accumulating pass over dx.  MTE_FOLD_SHORTCUT=0: two launches
        "fold_shortcut_dgrad": os.environ.get("MTE_FOLD_SHORTCUT", "1") == "1",
        # residual blocks: conv2's GroupNorm + ELU applied inside the kernel that forms x_out + Dropout2d(shortcut) and takes the sum's statistics
        # (ConvResidualTailFn / mte_gn_tail_fwd).  MTE_FUSE_TAIL=0: the round-4 form (ConvGnEluFn + ResidualTailFn over two tensors)
        "fuse_residual_tail": os.environ.get("MTE_FUSE_TAIL", "1") == "1",
        # round 5: the LDS-patch forward kernels leave the GroupNorm statistics of their output as per-tile records (no statistics pass over y)
        "gn_stats_in_conv": os.environ.get("MTE_GN_IN_CONV", "1") == "1"}
        # (round 5 also built the first pass of a GroupNorm's BACKWARD inside the data-gradient launch that writes its output gradient, and the 1x1 shortcut's
        #  forward launch on the side stream: both measured neutral -- profiles/r05_gn_in_conv.txt, DESIGN 7 -- and were removed in round 6)



def use_pack_folding(flag):
    """Enable/disable folding conv3d into the pack convolution (tests compare both formulations)."""
    _cfg["pack_folding"] = bool(flag)


def pack_folding_enabled():
    return _cfg["pack_folding"]


def use_patch_kernels(flag):
    """Enable/disable the LDS-patch conv kernels (tests compare them against the generic implicit GEMM)."""
    _cfg["patch_kernels"] = bool(flag)


def _stem_ok(W, cin_p, n, kh, kw, dtype):
    """the 8-input-channel stem kernels (csrc/conv_stem.hip) cover this shape"""
    return (_cfg["patch_kernels"] and cin_p == 8 and dtype == torch.bfloat16
            and lib.mte_conv2d_stem_supported(W, cin_p, n, kh, kw, DT_BF16) == 1)


def _patch_ok(W, cin_p, n, kh, kw, dtype):
    return _cfg["patch_kernels"] and dtype == torch.bfloat16 and lib.mte_conv2d_patch_supported(W, cin_p, n, kh, kw, DT_BF16) == 1


class SiteList:
    """Active set of a sparse NHWC map as a list of pixel indices (SAN branch): `rows` int32 [B*H*W] (the first `count[0]` entries are
    valid, raster order), `count` int32 [1] -- both on the device; the count never visits the host (mte_sparse_site_list)."""

    def __init__(self, mask):
        _require_gpu(mask)
        if mask.dtype != torch.uint8 or not mask.is_contiguous():
            raise MteError("site list: expected a contiguous uint8 mask [B,H,W]")
        n = mask.numel()
        self.rows = torch.empty((n,), dtype=torch.int32, device=mask.device)
        self.count = torch.empty((1,), dtype=torch.int32, device=mask.device)
        ws = torch.empty((int(lib.mte_sparse_site_list_workspace_elems(n)),), dtype=torch.int32, device=mask.device)
        lib.mte_sparse_site_list(mask.data_ptr(), n, self.rows.data_ptr(), self.count.data_ptr(), ws.data_ptr(), _stream())
        self.shape = tuple(mask.shape)


def conv_forward(x, wf, bias, cout, kh, kw, out=None, pack=None, w=None, accumulate=False, sites=None, gn_records=None):
    """y = conv_k(zero_pad(x)) + bias -> NHWC activation (written into `out` when given).
    sites (SiteList): the sparse form -- only the active sites are computed and written (mte_conv2d_igemm_sparse).
    gn_records (a list): the caller normalises y with GroupNorm(16) next -- where the launch can, it leaves the statistics of what it stores as per-tile
    records and appends (records, tiles per sample) to the list (round 5: mte_conv2d_patch_fwd_gn); an empty list afterwards = run the statistics pass."""
    B, Cp, H, W = x.shape
    if out is None:
        out = new_act(B, cout, H, W, x.dtype, x.device)
        if sites is not None:
            out.zero_()                                      # the sparse form writes the active sites only: the rest must not be uninitialised memory (NaN + 0 gradients downstream)
    xp, ldx = _pl(x)
    yp, ldy = _pl(out)
    if sites is not None:
        if sites.shape != (B, H, W):
            raise MteError("site list of a %s mask used on a %s map" % (sites.shape, (B, H, W)))
        lib.mte_conv2d_igemm_sparse(xp, ldx, wf.data_ptr(), _ptr(bias), yp, ldy, B, H, W, Cp, cout, kh, kw, _dt(x),
                                    sites.rows.data_ptr(), sites.count.data_ptr(), 1 if accumulate else 0, _stream())
        return out
    if not accumulate and _stem_ok(W, Cp, cout, kh, kw, x.dtype):
        lib.mte_conv2d_stem_fwd(xp, ldx, wf.data_ptr(), _ptr(bias), yp, ldy, B, H, W, cout, kh, kw, _stream())
        return out
    if pack is not None and _patch_ok(W, Cp, cout, kh, kw, x.dtype):
        # (not where the norm that follows holds its (sample, group) slabs on chip and takes the statistics itself -- the single-pass slab / cluster kernels of
        #  the low-resolution layers would ignore the records: round-5 advisor)
        if (gn_records is not None and _cfg["gn_stats_in_conv"] and cout % 16 == 0
                and lib.mte_gn_fwd_is_single_pass_b(B, H * W, cout, 0, _dt(x)) != 1):
            n = int(lib.mte_conv2d_patch_fwd_gn_elems(B, H, W))
            rec = torch.empty((n,), dtype=torch.float32, device=x.device)
            tiles = ctypes.c_int(0)
            lib.mte_conv2d_patch_fwd_gn(xp, ldx, pack.get_patch(w, 'f').data_ptr(), _ptr(bias), yp, ldy, B, H, W, Cp, cout, kh, kw,
                                        1 if accumulate else 0, rec.data_ptr(), n, ctypes.byref(tiles), _stream())
            gn_records.append((rec, tiles.value))
            return out
        lib.mte_conv2d_patch_fwd(xp, ldx, pack.get_patch(w, 'f').data_ptr(), _ptr(bias), yp, ldy, B, H, W, Cp, cout, kh, kw,
                                 1 if accumulate else 0, _stream())
        return out
    ws, ws_n = _splitk_workspace(B * H * W, cout, x.device)
    lib.mte_conv2d_igemm(xp, ldx, wf.data_ptr(), _ptr(bias), yp, ldy, 0, B, H, W, Cp, cout, kh, kw, _dt(x), _ptr(ws), ws_n,
                         (1 if accumulate else 0) | _CONV_SOLO, _stream())
    return out


_CONV_SOLO = 2      # MTE_CONV_SOLO: the forward pass has no weight-gradient kernels running beside it (see include/mte_kernels.h)


# ---- weight-gradient side stream --------------------------------------------------------------------------------
# dW of a layer depends only on (x, dy); the backward chain continues through dx.  When the gradient goes straight into
# the flat gradient buffer (GradSink) nothing on the main stream consumes it before the optimizer, so the wgrad kernels
# (+ their staging memset / OIHW unpack / bias column sums) are queued on a second HIP stream and fill the CUs that the
# many short, latency-bound kernels of the main chain (GroupNorm passes, low-resolution layers) leave idle.  The main
# stream re-joins at the end of the autograd pass (engine callback) and wherever gradients are consumed earlier
# (bucketed all-reduce).
_side = {"enabled": True, "streams": [], "next": 0, "keep": [], "keep_bytes": 0, "callback_queued": False, "dirty": False}
lib.set_option(3, 1, lazy=True)
def _new_side_stream():
    """the weight-gradient queue, one HIP priority level BELOW the main queue: the step waits for the main queue, whose one-workgroup-per-CU kernels lose a whole
    second round of workgroups to every CU the side queue holds.  Same box, alternating, 4 x 30 steps (profiles/r06_side_priority.txt): equal priority 21.60-21.69 ms,
    below 21.49-21.61, above 21.70.  MTE_SIDE_PRIORITY (development): 0 = equal, -1 = above."""
    pr = int(os.environ.get("MTE_SIDE_PRIORITY", "1"))
    if pr == 0:
        return torch.cuda.Stream()
    hip = ctypes.CDLL("libamdhip64.so")
    h = ctypes.c_void_p()
    if hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, pr) != 0:           # 1 = hipStreamNonBlocking
        raise RuntimeError("hipStreamCreateWithPriority(%d) failed" % pr)
    return torch.cuda.ExternalStream(h.value)


_SIDE_STREAMS = 1      # (round 5 measured two weight-gradient streams no faster than one: profiles/r05_side_queue_width.txt; the option is gone)
_SIDE_KEEP_LIMIT = 24 << 30          # bytes of (x, dy) kept alive for the side stream before a forced join


def use_wgrad_side_stream(flag):
    _side["enabled"] = bool(flag)
    lib.set_option(3, 1 if flag else 0, lazy=True)           # MTE_OPT_WGRAD_SHARES_CHIP: the weight-gradient launch widths follow the schedule


class _BandChain:
    """The exact-border band chain of a folded pack layer (thin strips: 16 x 5 x 640 pixels and the like).  Rounds 3-5 could run it on a third stream beside the
    layer's full-size kernels (MTE_BAND_STREAM=1: +0.4 % on one GPU, but 25.4 -> 35.1 ms per step as soon as RCCL's streams exist and the third stream shares a
    hardware queue with one of the other two -- profiles/r03_band_stream_ab.txt); it lost that A/B and the option was removed in round 6.  What is left is the
    bracket around the chain in PackFoldedConvGnEluFn: the chain runs on the current stream, `mark` / `join` are no-ops."""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def mark(self):
        return None

    def join(self, ev=None):
        return None


def join_side_stream():
    """Make the current stream wait for every weight-gradient kernel queued on the side stream so far."""
    if _side["dirty"]:
        for st in _side["streams"]:
            ev = torch.cuda.Event()
            ev.record(st)
            torch.cuda.current_stream().wait_event(ev)
        _side["dirty"] = False
    _side["keep"].clear()
    _side["keep_bytes"] = 0


def side_streams_wait_into(stream):
    """Make ``stream`` wait for the weight-gradient kernels queued so far WITHOUT joining them into the current stream
    (used by the gradient all-reduce, which must not stall the rest of backward)."""
    if _side["dirty"]:
        for st in _side["streams"]:
            ev = torch.cuda.Event()
            ev.record(st)
            stream.wait_event(ev)


def _side_join_callback():
    _side["callback_queued"] = False
    join_side_stream()


def _side_stream_for(*tensors):
    """-> side stream (after making it wait for the work queued so far on the current stream), keeping `tensors` alive
    until the next join (their memory must not be recycled by main-stream allocations while the side stream reads it)."""
    if not _side["streams"]:
        _side["streams"] = [_new_side_stream() for _ in range(max(1, _SIDE_STREAMS))]
    side = _side["streams"][_side["next"] % len(_side["streams"])]
    _side["next"] += 1
    ev = torch.cuda.Event()
    ev.record()
    side.wait_event(ev)
    _side["keep"].append(tensors)
    _side["keep_bytes"] += sum(t.numel() * t.element_size() for t in tensors)
    _side["dirty"] = True
    if not _side["callback_queued"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_side_join_callback)
            _side["callback_queued"] = True
        except RuntimeError:          # not inside a backward pass: the caller joins explicitly
            pass
    return side


def _conv_wgrad(x, dy, w, need_dbias, dw_out, dbias_out):
    cout, cin, kh, kw = w.shape
    B, Cp, H, W = x.shape
    dyp, lddy = _pl(dy)
    xp, ldx = _pl(x)
    st = _stream()
    dbias = None
    # room for one partial gradient per pixel split (plain stores, summed by the unpack pass; +0.5 % over fp32 atomics for the
    # generic kernel's <= 32 splits -- the patch kernel's 128..512 workgroup groups stay on atomics: summing that many parts
    # in the unpack pass measured 7 % slower)
    per = cout * kh * kw * Cp
    patch = _cfg["patch_kernels"] and x.dtype == torch.bfloat16 and lib.mte_conv2d_patch_wgrad_supported(W, Cp, cout, kh, kw, DT_BF16) == 1
    if patch and lib.mte_conv2d_wgrad_nine_tap(H, W, Cp, cout, kh, kw, _dt(x)) == 1:
        patch = False                    # 3x3 with 128 outputs: the nine-tap kernel (round 5: 128 -> 128 @96x320 100.5 -> 84.9 us, 192 -> 128 147 -> 115 incl. the unpack pass)
    if kh * kw == 1 and Cp >= 64:
        patch = False                    # 1x1 with >= 64 inputs: the generic kernel with many pixel splits streams both operands at 4.5 TB/s (the patch kernel deals TAPS to its waves: 91 -> 56 us)
    # the LDS-patch kernel runs 128..512 workgroups per layer: one slab each (plain stores), combined by a two-level reduction
    # (mte_unpack_conv_wgrad) -- its slabs are 70-210 KB, so even 512 of them stay near 100 MB
    # generic kernel: <= 32 pixel splits -- except where a slab is small (1x1 shortcuts: 64 KB .. 1 MB): one output tile means the pixel splits
    # ARE the launch's workgroups, and 32 of them streamed the two 63 MB operands of a full-resolution shortcut at 1.4 TB/s (round 4)
    # (0.091 -> 0.035 ms for 128 -> 128 at 96x320, 0.051 -> 0.026 ms for 256 -> 256 at 48x160 with up to 256 splits; the two-level part sum takes > 32 parts)
    wide = 256 if per <= (1 << 18) else (64 if per <= (1 << 19) else 32)      # (64: the nine-tap kernel's 128 -> 256 layer, 4 tiles x 64 pixel splits)
    cap = max(1, min(512, (192 << 20) // (4 * per))) if patch and _cfg["patch_wgrad_slabs"] else max(1, min(wide, (96 << 20) // (4 * per)))
    stem = _stem_ok(W, Cp, cout, kh, kw, x.dtype)
    if stem:
        cap = 2048                       # 25 KB slabs: eight 256-thread workgroups per CU keep this bandwidth-bound stream busy
    stage = torch.empty((cap + (32 if cap > 32 else 0), cout, kh * kw, Cp), dtype=torch.float32, device=x.device)   # (+ 32: scratch of the two-level part sum)
    parts = ctypes.c_int(1)
    if stem:
        lib.mte_conv2d_stem_wgrad(xp, ldx, dyp, lddy, stage.data_ptr(), cap, ctypes.byref(parts), B, H, W, cout, kh, kw, st)
    elif patch:
        lib.mte_conv2d_patch_wgrad(xp, ldx, dyp, lddy, stage.data_ptr(), cap, ctypes.byref(parts), B, H, W, Cp, cout, kh, kw, st)
    else:
        lib.mte_conv2d_wgrad(xp, ldx, dyp, lddy, stage.data_ptr(), cap, ctypes.byref(parts), B, H, W, Cp, cout, kh, kw, _dt(x), st)
    dw = dw_out if dw_out is not None else torch.empty_like(w, dtype=torch.float32)
    lib.mte_unpack_conv_wgrad(stage.data_ptr(), parts.value, dw.data_ptr(), cout, cin, kh, kw, Cp, st)
    if need_dbias:
        dbias = dbias_out if dbias_out is not None else torch.empty((cout,), dtype=torch.float32, device=x.device)
        lib.mte_colsum(dyp, lddy, B * H * W, cout, dbias.data_ptr(), _dt(dy), st)
    return dw, dbias


def conv_backward(x, dy, w, pack, need_dx, need_dw=True, need_dbias=True, dw_out=None, dbias_out=None, fork_slot=None, sunk=False, sites=None):
    """-> (dx or None, dw (OIHW fp32) or None, dbias fp32); dw_out / dbias_out: pre-allocated destinations; sunk: they are views
    of the gradient sink (nothing on the backward chain reads them: the weight-gradient kernels may run on the side stream);
    fork_slot: see ForkFn -- the data gradient is accumulated into the gradient another consumer of x already produced"""
    dw = dbias = dx = None
    on_side = False
    if need_dw:
        sunk = bool(sunk) and dw_out is not None and (dbias_out is not None or not need_dbias)
        on_side = sunk and _side["enabled"] and need_dx

    def wgrad():
        if on_side:
            if _side["keep_bytes"] > _SIDE_KEEP_LIMIT:
                join_side_stream()
            with torch.cuda.stream(_side_stream_for(x, dy)):
                return _conv_wgrad(x, dy, w, need_dbias, dw_out, dbias_out)
        return _conv_wgrad(x, dy, w, need_dbias, dw_out, dbias_out)

    # (Round 6 measured the other order -- the weight gradient forked BEHIND the data-gradient launch, so that it starts beside the previous layer's memory-bound
    #  GroupNorm backward instead of beside the matrix-bound data gradient: 21.74 / 21.82 / 21.66 ms against 21.75 / 21.69 / 21.87, no difference:
    #  profiles/r06_wgrad_after_dgrad.txt.  The side queue is a FIFO that runs behind the main queue anyway; where a launch is forked does not place it.)
    if need_dw:
        dw, dbias = wgrad()
    if need_dx:
        dx = _conv_dgrad(x, dy, w, pack, fork_slot, sites)
    return dx, dw, dbias


def _conv_dgrad(x, dy, w, pack, fork_slot, sites):
    """the data gradient of conv_backward (see there)"""
    cout, cin, kh, kw = w.shape
    B, Cp, H, W = x.shape
    dyp, lddy = _pl(dy)
    st = _stream()
    target = _fork_target(fork_slot, (B, Cp, H, W), x.dtype, keep_pending=True)
    if (target is None and fork_slot is not None and (kh, kw) == (1, 1) and sites is None and _cfg["fold_shortcut_dgrad"]
            and not _cfg.get("no_fork_accumulate") and _patch_ok(W, cout, Cp, 1, 1, x.dtype) and _patch_ok(W, 32, Cp, 3, 3, x.dtype)):
        # a 1x1 whose input has another consumer that has not produced its data gradient yet (a residual block's shortcut: autograd runs it before
        # the block's conv1): hand out the buffer, leave the launch to that consumer's 3x3 data gradient (below) -- or to whoever touches the slot first
        dx = new_act(B, Cp, H, W, x.dtype, x.device)
        fork_slot["buf"] = dx
        fork_slot["pending"] = (dy, w, pack)
        return dx
    pend = _pending_slot(fork_slot) if target is not None else None
    if pend is not None:
        dy3, w3, pack3 = pend["pending"]
        if ((kh, kw) == (3, 3) and sites is None and target is pend["buf"] and _patch_ok(W, cout, Cp, 3, 3, x.dtype)
                and tuple(dy3.shape[2:]) == (H, W) and dy3.dtype == x.dtype):
            del pend["pending"]
            d3p, ldd3 = _pl(dy3)
            dxp, lddx = _pl(target)
            lib.mte_conv2d_patch_fwd_plus1x1(dyp, lddy, pack.get_patch(w, 'b').data_ptr(), 0, dxp, lddx, B, H, W, cout, Cp,
                                             d3p, ldd3, pack3.get_patch(w3, 'b').data_ptr(), dy3.shape[1], st)
            fork_slot["buf"] = target
            return target
        _flush_pending(pend)
    dx = target if target is not None else new_act(B, Cp, H, W, x.dtype, x.device)
    if target is None and sites is not None:
        dx.zero_()                                       # (as in conv_forward: inactive sites of a sparse result are exact zeros)
    acc = 1 if target is not None else 0
    dxp, lddx = _pl(dx)
    if sites is not None:                              # sparse data gradient: the same gather-GEMM-scatter with the backward pack
        _, wb = pack.get(w, x.dtype, True)
        lib.mte_conv2d_igemm_sparse(dyp, lddy, wb.data_ptr(), 0, dxp, lddx, B, H, W, cout, Cp, kh, kw, _dt(x),
                                    sites.rows.data_ptr(), sites.count.data_ptr(), acc, st)
    elif _patch_ok(W, cout, Cp, kh, kw, x.dtype):
        lib.mte_conv2d_patch_fwd(dyp, lddy, pack.get_patch(w, 'b').data_ptr(), 0, dxp, lddx, B, H, W, cout, Cp, kh, kw, acc, st)
    else:
        _, wb = pack.get(w, x.dtype, True)
        ws, ws_n = _splitk_workspace(B * H * W, Cp, x.device)
        lib.mte_conv2d_igemm(dyp, lddy, wb.data_ptr(), 0, dxp, lddx, 0, B, H, W, cout, Cp, kh, kw, _dt(x), _ptr(ws), ws_n, acc, st)
    if fork_slot is not None:
        fork_slot["buf"] = dx
    return dx


_stats_elems = {}


def gn_stats_buffer(B, device):
    """GroupNorm statistics buffer of a batch (include/mte_kernels.h, mte_gn_stats_elems): [B][16][2] sums that the normalisation
    and its backward read, followed by the statistics pass's arrival tickets (must be zero: the arena is) and per-workgroup records."""
    n = _stats_elems.get(B)
    if n is None:
        n = _stats_elems[B] = int(lib.mte_gn_stats_elems(B))
    return _zeros((n,), torch.float64, device)


def _gn_forward(y1, y2, scale2, gamma, beta, eps, out=None, records=None):
    """records: [(per-tile records, tiles per sample)] left by the convolution that produced y1 (conv_forward(gn_records=...)) -- the statistics pass
    over y1 is replaced by the sum of the records"""
    B, C, H, W = y1.shape
    p1, l1 = _pl(y1)
    p2, l2 = _pl(y2) if y2 is not None else (0, 0)
    st = _stream()
    ready = 1
    stats = gn_stats_buffer(B, y1.device)
    if lib.mte_gn_fwd_is_single_pass_b(B, H * W, C, 1 if y2 is not None else 0, _dt(y1)) == 1:
        ready = 0                      # one kernel holds each (sample, group) slab on chip (one workgroup, or a cluster of them) -- statistics + apply
    elif records and y2 is None:
        lib.mte_gn_stats_from_records(records[0][0].data_ptr(), records[0][1], stats.data_ptr(), B, st)
    else:
        lib.mte_gn_stats(p1, l1, p2, l2, _ptr(scale2), stats.data_ptr(), B, H * W, C, _dt(y1), st)
    z = out if out is not None else new_act(B, C, H, W, y1.dtype, y1.device)
    if tuple(z.shape) != (B, C, H, W) or z.dtype != y1.dtype:
        raise MteError("output destination has shape %s / %s, expected %s / %s" % (tuple(z.shape), z.dtype, (B, C, H, W), y1.dtype))
    zp, lz = _pl(z)
    lib.mte_gn_elu_fwd(p1, l1, p2, l2, _ptr(scale2), stats.data_ptr(), ready, gamma.data_ptr(), beta.data_ptr(), zp, lz,
                       B, H * W, C, eps, _dt(y1), st)
    return z, stats


def _gn_backward(dz, y1, y2, scale2, stats, gamma, beta, eps, need_d2, want_dbias=False, dgamma=None, dbeta=None, dbias=None):
    B, C, H, W = y1.shape
    dz = as_act(dz, y1.dtype)
    red = _zeros((B, C, 2), torch.float32, y1.device)
    d1 = new_act(B, C, H, W, y1.dtype, y1.device)
    d2 = new_act(B, C, H, W, y1.dtype, y1.device) if need_d2 else None
    # zero at entry (MTE_OPT_GN_PREZEROED covers dgamma / dbeta too: the single-pass kernels ADD per-sample parts into them)
    dgamma = dgamma if dgamma is not None else _zeros((C,), torch.float32, y1.device)
    dbeta = dbeta if dbeta is not None else _zeros((C,), torch.float32, y1.device)
    if want_dbias and dbias is None:
        dbias = _zeros((C,), torch.float32, y1.device)
    pz, lz = _pl(dz)
    p1, l1 = _pl(y1)
    p2, l2 = _pl(y2) if y2 is not None else (0, 0)
    pd1, ld1 = _pl(d1)
    pd2, ld2 = _pl(d2) if d2 is not None else (0, 0)
    lib.mte_gn_elu_bwd(pz, lz, p1, l1, p2, l2, _ptr(scale2), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), red.data_ptr(),
                       pd1, ld1, pd2, ld2, dgamma.data_ptr(), dbeta.data_ptr(), _ptr(dbias), B, H * W, C, eps, _dt(y1), _stream())
    if want_dbias:
        return d1, d2, dgamma, dbeta, dbias
    return d1, d2, dgamma, dbeta


GN_EPS = 1e-5

_dropout_pool = {}


def dropout2d_scale(B, C, p, device):
    """Dropout2d channel factors keep/(1-p) for one layer, [B, C] fp32 (reference: nn.Dropout2d in ResidualConv.conv3,
    layers01.py:58-61).  The Bernoulli draws of a whole step's layers come from one pooled torch.rand (4 launches per
    refill instead of 4 per layer); draws are i.i.d., so pooling does not change the distribution."""
    key = (str(device), float(p))
    pool = _dropout_pool.get(key)
    n = B * C
    if pool is None or pool[1] + n > pool[0].numel():
        size = max(1 << 17, 2 * n)
        pool = _dropout_pool[key] = [(torch.rand((size,), device=device) >= p).float() / (1.0 - p), 0]
    out = pool[0][pool[1]:pool[1] + n].view(B, C)
    pool[1] += n
    return out



class ConvGnEluFn(torch.autograd.Function):
    """ELU(GroupNorm16(conv_k(zero_pad(x)) + b))  -- reference Conv2D.forward, layers01.py:35-38."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, pack, out=None):
        ctx.fork_slot = getattr(x, "_mte_fork_slot", None)
        wf, _ = pack.get(w, x.dtype, bool(ctx.needs_input_grad[0]))
        cout, cin, kh, kw = w.shape
        recs = []
        y = conv_forward(x, wf, b, cout, kh, kw, pack=pack, w=w, gn_records=recs)
        z, stats = _gn_forward(y, None, None, gamma, beta, GN_EPS, out=None if out is None else alias_of(out), records=recs)
        ctx.save_for_backward(x, w, y, stats, gamma, beta)
        ctx.pack = pack
        ctx.bias = b
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w, y, stats, gamma, beta = ctx.saved_tensors
        b = ctx.bias
        gg, sg = _grad_dst(gamma, zero=True)
        gb, sb = _grad_dst(beta, zero=True)
        gbias, sbias = _grad_dst(b, zero=True)
        gw, sw = _grad_dst(w)
        dy, _, dgamma, dbeta, db = _gn_backward(dz, y, None, None, stats, gamma, beta, GN_EPS, False, want_dbias=True,
                                                dgamma=gg, dbeta=gb, dbias=gbias)
        dx, dw, _ = conv_backward(x, dy, w, ctx.pack, ctx.needs_input_grad[0], need_dbias=False, dw_out=gw, fork_slot=ctx.fork_slot, sunk=sw)
        return dx, _grad_ret(w, dw, sw), _grad_ret(b, db, sbias), _grad_ret(gamma, dgamma, sg), _grad_ret(beta, dbeta, sb), None, None


def _main_weight(pack, w, cm):
    """contiguous copy of w[:, :cm] (the channels the GEMM kernels see), refreshed in place when the parameter changed -- the weight packs key on
    this tensor's version, so a refresh re-packs and nothing else does"""
    key = (w.data_ptr(), w._version, weights_epoch())
    wm = getattr(pack, "_main_w", None)
    if wm is None or tuple(wm.shape) != (w.shape[0], cm, w.shape[2], w.shape[3]) or wm.device != w.device:
        wm = pack._main_w = torch.empty((w.shape[0], cm, w.shape[2], w.shape[3]), dtype=torch.float32, device=w.device)
        pack._main_key = None
    pack._main_src = (weakref.ref(w), cm)                # prefetch_weight_packs refreshes wm from the parameter before it re-packs (side stream)
    if pack._main_key != key:
        wm.copy_(w.detach()[:, :cm])
        pack._main_key = key
    return wm


def _refresh_main_weights(packs):
    """called by prefetch_weight_packs on the stream that re-packs, after the optimizer changed the parameters: the contiguous w[:, :C-1] copies of the
    rank-1 iconv layers (ConvGnEluInvFn keys its packs on the COPY) are brought up to date first -- round 4 re-packed them from the stale copy and the
    next forward pass then refreshed copy and packs on the main stream without waiting for that launch (advisor: a cross-stream write-after-read)."""
    for pk in packs:
        src = getattr(pk, "_main_src", None)
        wm = getattr(pk, "_main_w", None)
        if src is None or wm is None:
            continue
        w = src[0]()
        if w is None or tuple(wm.shape) != (w.shape[0], src[1], w.shape[2], w.shape[3]):
            continue
        key = (w.data_ptr(), w._version, weights_epoch())
        if pk._main_key != key:
            wm.copy_(w.detach()[:, :src[1]])
            pk._main_key = key


class ConvGnEluInvFn(torch.autograd.Function):
    """ELU(GroupNorm16(conv_3(zero_pad(cat(x, nearest_up2(inv)))) + b)) -- the decoder's iconv3 / iconv2 / iconv1 (reference PackNetSAN01.py:118-143 +
    Conv2D.forward) with the inverse-depth channel as a rank-1 term: mte_rank1_conv_fwd writes conv_1(up2(inv)) into y, the GEMM kernels accumulate
    the convolution of the other C - 1 channels (x) onto it.  inv: [B,1,H/2,W/2] fp32 (the InvDepth head's output)."""

    @staticmethod
    def forward(ctx, x, inv, w, b, gamma, beta, pack, out=None):
        ctx.fork_slot = getattr(x, "_mte_fork_slot", None)
        cout, cin, kh, kw = w.shape
        cm = cin - 1
        B, Cp, H, W = x.shape
        if (kh, kw) != (3, 3) or Cp != cm or tuple(inv.shape) != (B, 1, H // 2, W // 2):
            raise MteError("rank-1 inverse-depth term: expected a 3x3 layer over %d + 1 channels and a [B,1,H/2,W/2] map, got %s / %s / %s"
                           % (Cp, tuple(w.shape), tuple(x.shape), tuple(inv.shape)))
        wm = _main_weight(pack, w, cm)
        wf, _ = pack.get(wm, x.dtype, bool(ctx.needs_input_grad[0]))
        invc = inv.detach().contiguous().float()
        y = new_act(B, cout, H, W, x.dtype, x.device)
        yp, ldy = _pl(y)
        w1 = w.detach().data_ptr() + 4 * cm * 9
        xp, ldx = _pl(x)
        if (_cfg["fuse_inv_term"] and pack is not None and _patch_ok(W, Cp, cout, kh, kw, x.dtype)
                and lib.mte_conv2d_patch_fwd_rank1_ok(_ptr(b), ldx, B, H, W, Cp, cout) == 1):
            # one launch: the tile's store loop adds the map's term (no pass over y before the conv, no read of y in it)
            lib.mte_conv2d_patch_fwd_rank1(xp, ldx, pack.get_patch(wm, 'f').data_ptr(), _ptr(b), yp, ldy, B, H, W, Cp, cout,
                                           invc.data_ptr(), w1, cin * 9, _stream())
        else:
            lib.mte_rank1_conv_fwd(invc.data_ptr(), w1, cin * 9, yp, ldy, B, H // 2, W // 2, cout, _dt(y), _stream())
            conv_forward(x, wf, b, cout, kh, kw, out=y, pack=pack, w=wm, accumulate=True)
        z, stats = _gn_forward(y, None, None, gamma, beta, GN_EPS, out=None if out is None else alias_of(out))
        ctx.save_for_backward(x, invc, w, wm, y, stats, gamma, beta)
        ctx.pack = pack
        ctx.bias = b
        return z

    @staticmethod
    def backward(ctx, dz):
        x, invc, w, wm, y, stats, gamma, beta = ctx.saved_tensors
        b = ctx.bias
        cout, cin, kh, kw = w.shape
        cm = cin - 1
        B, Cp, H, W = x.shape
        gg, sg = _grad_dst(gamma, zero=True)
        gb, sb = _grad_dst(beta, zero=True)
        gbias, sbias = _grad_dst(b, zero=True)
        gw, sw = _grad_dst(w)
        dy, _, dgamma, dbeta, db = _gn_backward(dz, y, None, None, stats, gamma, beta, GN_EPS, False, want_dbias=True,
                                                dgamma=gg, dbeta=gb, dbias=gbias)
        dyp, lddy = _pl(dy)

        # channel C-1 (the map): its data gradient on the main stream (it feeds the head's backward) ...
        dinv = torch.empty((B, 1, H // 2, W // 2), dtype=torch.float32, device=x.device)
        wl = w.detach()
        lib.mte_rank1_conv_bwd_data(dyp, lddy, wl.data_ptr() + 4 * cm * 9, cin * 9, dinv.data_ptr(), B, H // 2, W // 2, cout, 0, _dt(dy), _stream())

        def weight_gradients():
            # ... channels 0 .. C-2: the GEMM weight gradient over x; channel C-1: one pass over dy against the map, straight into column C-1 of gw
            dwm, _ = _conv_wgrad(x, dy, wm, False, None, None)
            gw[:, :cm].copy_(dwm)
            rec = torch.empty((int(lib.mte_rank1_conv_bwd_records_elems(cout)),), dtype=torch.float32, device=x.device)
            lib.mte_rank1_conv_bwd_weight(dyp, lddy, invc.data_ptr(), gw.data_ptr() + 4 * cm * 9, cin * 9, rec.data_ptr(), B, H // 2, W // 2, cout, _dt(dy), _stream())

        if sw and _side["enabled"]:
            if _side["keep_bytes"] > _SIDE_KEEP_LIMIT:
                join_side_stream()
            with torch.cuda.stream(_side_stream_for(x, dy, invc)):
                weight_gradients()
        else:
            weight_gradients()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv_backward(x, dy, wm, ctx.pack, True, need_dw=False, need_dbias=False, fork_slot=ctx.fork_slot)[0]
        if not ctx.needs_input_grad[1]:
            dinv = None
        return dx, dinv, _grad_ret(w, gw, sw), _grad_ret(b, db, sbias), _grad_ret(gamma, dgamma, sg), _grad_ret(beta, dbeta, sb), None, None


class ConvFn(torch.autograd.Function):
    """Plain conv + bias (the 1x1 shortcut of ResidualConv, layers01.py:61)."""

    @staticmethod
    def forward(ctx, x, w, b, pack, bias_grad=True):
        """bias_grad=False: the consumer produces the bias gradient (ResidualTailFn: the column sums of this conv's output gradient fall
        out of the GroupNorm backward pass that computes it -- no stand-alone column-sum launch)."""
        ctx.fork_slot = getattr(x, "_mte_fork_slot", None)
        wf, _ = pack.get(w, x.dtype, bool(ctx.needs_input_grad[0]))
        cout, cin, kh, kw = w.shape
        y = conv_forward(x, wf, b, cout, kh, kw, pack=pack, w=w)
        ctx.save_for_backward(x, w)
        ctx.pack = pack
        ctx.bias = b
        ctx.bias_grad = bool(bias_grad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        b = ctx.bias
        dy = as_act(dy, x.dtype)
        gw, sw = _grad_dst(w)
        if not ctx.bias_grad:
            dx, dw, _ = conv_backward(x, dy, w, ctx.pack, ctx.needs_input_grad[0], need_dbias=False, dw_out=gw, fork_slot=ctx.fork_slot, sunk=sw)
            return dx, _grad_ret(w, dw, sw), None, None, None
        gbias, sbias = _grad_dst(b)
        dx, dw, db = conv_backward(x, dy, w, ctx.pack, ctx.needs_input_grad[0], dw_out=gw, dbias_out=gbias, fork_slot=ctx.fork_slot,
                                   sunk=sw and sbias)
        return dx, _grad_ret(w, dw, sw), _grad_ret(b, db, sbias), None, None


class ResidualTailFn(torch.autograd.Function):
    """ELU(GroupNorm16(a + scale[b,c] * s)) -- ResidualConv tail with Dropout2d folded in (layers01.py:65-73)."""

    @staticmethod
    def forward(ctx, a, s, scale, gamma, beta, bias_s=None):
        """bias_s: the bias parameter of the conv that produced `s` (the block's 1x1 shortcut).  Its value is already inside `s`; it is an
        input here only so that its GRADIENT -- sum of ds over batch and pixels -- can come out of this op's backward pass, where ds is
        formed anyway (the shortcut conv is then built with bias_grad=False)."""
        z, stats = _gn_forward(a, s, scale, gamma, beta, GN_EPS)
        ctx.save_for_backward(a, s, stats, gamma, beta)
        ctx.scale = scale
        ctx.bias_s = bias_s
        return z

    @staticmethod
    def backward(ctx, dz):
        a, s, stats, gamma, beta = ctx.saved_tensors
        gg, sg = _grad_dst(gamma, zero=True)
        gb, sb = _grad_dst(beta, zero=True)
        if ctx.bias_s is not None and ctx.needs_input_grad[5]:
            gbs, sbs = _grad_dst(ctx.bias_s, zero=True)
            da, ds, dgamma, dbeta, dbs = _gn_backward(dz, a, s, ctx.scale, stats, gamma, beta, GN_EPS, True, want_dbias=True,
                                                      dgamma=gg, dbeta=gb, dbias=gbs)
            return da, ds, None, _grad_ret(gamma, dgamma, sg), _grad_ret(beta, dbeta, sb), _grad_ret(ctx.bias_s, dbs, sbs)
        da, ds, dgamma, dbeta = _gn_backward(dz, a, s, ctx.scale, stats, gamma, beta, GN_EPS, True, dgamma=gg, dbeta=gb)
        return da, ds, None, _grad_ret(gamma, dgamma, sg), _grad_ret(beta, dbeta, sb), None


def residual_tail_fused_ok(B, C, H, W, dtype):
    """the two-launch residual tail (ConvResidualTailFn) is the better form: everywhere except where the tail over two tensors is a single-pass cluster
    kernel already (512 channels at 24x80 with B = 8: one read of each input, one write)"""
    if not _cfg["fuse_residual_tail"]:
        return False
    return lib.mte_gn_fwd_is_single_pass_b(B, H * W, C, 1, DT_BF16 if dtype == torch.bfloat16 else DT_F32) != 1


class ConvResidualTailFn(torch.autograd.Function):
    """ELU(GN_t(ELU(GN_2(conv2(x1) + b2)) + scale[b,c] * s)) -- conv2 of a ResidualConv TOGETHER with the block's tail (reference layers01.py:59-73:
    `x_out = self.conv2(x_out)`, `self.activ(self.normalize(x_out + shortcut))`; s = the 1x1 shortcut, scale = Dropout2d's factor or None).
    Round 5: conv2's own GroupNorm + ELU is applied on the fly by the kernel that forms the sum t = x_out + scale * s, which also takes t's statistics
    (mte_gn_tail_fwd): x_out is never written, the stand-alone statistics pass over (x_out, s) is gone, and both backward passes of the outer norm read t
    instead of (x_out, s).  Saved for backward: conv2's convolution output c2 and t (before: c2, x_out and s)."""

    @staticmethod
    def forward(ctx, x1, w2, b2, gamma2, beta2, pack2, s, scale, gamma_t, beta_t, bias_s):
        ctx.fork_slot = getattr(x1, "_mte_fork_slot", None)
        wf, _ = pack2.get(w2, x1.dtype, bool(ctx.needs_input_grad[0]))
        cout, cin, kh, kw = w2.shape
        recs = []
        c2 = conv_forward(x1, wf, b2, cout, kh, kw, pack=pack2, w=w2, gn_records=recs)
        B, C, H, W = c2.shape
        if tuple(s.shape) != (B, C, H, W) or s.dtype != c2.dtype:
            raise MteError("residual tail: shortcut %s / %s against %s / %s" % (tuple(s.shape), s.dtype, (B, C, H, W), c2.dtype))
        st = _stream()
        p1, l1 = _pl(c2)
        p2, l2 = _pl(s)
        stats2 = gn_stats_buffer(B, c2.device)
        if recs:                       # (round 5) conv2 left its statistics as per-tile records
            lib.mte_gn_stats_from_records(recs[0][0].data_ptr(), recs[0][1], stats2.data_ptr(), B, st)
        else:
            lib.mte_gn_stats(p1, l1, 0, 0, 0, stats2.data_ptr(), B, H * W, C, _dt(c2), st)
        stats_t = gn_stats_buffer(B, c2.device)
        t = new_act(B, C, H, W, c2.dtype, c2.device)
        z = new_act(B, C, H, W, c2.dtype, c2.device)
        pt, lt = _pl(t)
        pz, lz = _pl(z)
        lib.mte_gn_tail_fwd(p1, l1, stats2.data_ptr(), gamma2.data_ptr(), beta2.data_ptr(), p2, l2, _ptr(scale), pt, lt, stats_t.data_ptr(),
                            gamma_t.data_ptr(), beta_t.data_ptr(), pz, lz, B, H * W, C, GN_EPS, _dt(c2), st)
        ctx.save_for_backward(x1, w2, c2, stats2, gamma2, beta2, t, stats_t, gamma_t, beta_t)
        ctx.scale = scale
        ctx.bias_s = bias_s
        ctx.bias2 = b2
        ctx.pack = pack2
        return z

    @staticmethod
    def backward(ctx, dz):
        x1, w2, c2, stats2, gamma2, beta2, t, stats_t, gamma_t, beta_t = ctx.saved_tensors
        scale, bias_s, b2 = ctx.scale, ctx.bias_s, ctx.bias2
        # ---- outer norm: dt (= the gradient of conv2's Conv2D output) and ds = scale * dt (the shortcut's output gradient; its bias gradient = column sums)
        ggt, sgt = _grad_dst(gamma_t, zero=True)
        gbt, sbt = _grad_dst(beta_t, zero=True)
        want_bs = bias_s is not None and ctx.needs_input_grad[10]
        gbs, sbs = _grad_dst(bias_s, zero=True) if want_bs else (None, False)
        if scale is None:          # no Dropout2d: ds IS dt
            out = _gn_backward(dz, t, None, None, stats_t, gamma_t, beta_t, GN_EPS, False, want_dbias=want_bs, dgamma=ggt, dbeta=gbt, dbias=gbs)
            dt = ds = out[0]
        else:
            out = _gn_backward(dz, t, None, scale, stats_t, gamma_t, beta_t, GN_EPS, True, want_dbias=want_bs, dgamma=ggt, dbeta=gbt, dbias=gbs)
            dt, ds = out[0], out[1]
        dgamma_t, dbeta_t = out[2], out[3]
        dbs = out[4] if want_bs else None
        # ---- conv2's own norm, then its convolution
        gg2, sg2 = _grad_dst(gamma2, zero=True)
        gb2, sb2 = _grad_dst(beta2, zero=True)
        gbias2, sbias2 = _grad_dst(b2, zero=True)
        gw2, sw2 = _grad_dst(w2)
        dc2, _, dgamma2, dbeta2, db2 = _gn_backward(dt, c2, None, None, stats2, gamma2, beta2, GN_EPS, False, want_dbias=True,
                                                    dgamma=gg2, dbeta=gb2, dbias=gbias2)
        dx1, dw2, _ = conv_backward(x1, dc2, w2, ctx.pack, ctx.needs_input_grad[0], need_dbias=False, dw_out=gw2, fork_slot=ctx.fork_slot, sunk=sw2)
        return (dx1, _grad_ret(w2, dw2, sw2), _grad_ret(b2, db2, sbias2), _grad_ret(gamma2, dgamma2, sg2), _grad_ret(beta2, dbeta2, sb2), None,
                ds if ctx.needs_input_grad[6] else None, None, _grad_ret(gamma_t, dgamma_t, sgt), _grad_ret(beta_t, dbeta_t, sbt),
                _grad_ret(bias_s, dbs, sbs) if want_bs else None)


class Pack3dFn(torch.autograd.Function):
    """packing(r=2) + Conv3d(1->4) + view  (PackLayerConv3d.forward before its Conv2D, layers01.py:241-246)."""

    @staticmethod
    def forward(ctx, x, w3, b3):
        B, C, H, W = x.shape
        out = new_act(B, 16 * C, H // 2, W // 2, x.dtype, x.device)
        xp, ldx = _pl(x)
        op, ldo = _pl(out)
        w3c, b3c = w3.detach().contiguous().float(), b3.detach().contiguous().float()
        lib.mte_pack3d_fwd(xp, ldx, w3c.data_ptr(), b3c.data_ptr(), op, ldo, B, H, W, C, _dt(x), _stream())
        ctx.save_for_backward(x, w3c)
        ctx.params = (w3, b3)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w3c = ctx.saved_tensors
        B, C, H, W = x.shape
        dout = as_act(dout, x.dtype)
        dop, ldo = _pl(dout)
        dw3, db3 = _conv3d_weight_grads(lib.mte_pack3d_bwd_weight, x, dout, *ctx.params)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new_act(B, C, H, W, x.dtype, x.device)
            dxp, lddx = _pl(dx)
            lib.mte_pack3d_bwd_data(dop, ldo, w3c.data_ptr(), dxp, lddx, B, H, W, C, _dt(x), _stream())
        return dx, dw3, db3


class Unpack3dFn(torch.autograd.Function):
    """Conv3d(1->4) + view + PixelShuffle(2)  (UnpackLayerConv3d.forward after its Conv2D, layers01.py:281-286)."""

    @staticmethod
    def forward(ctx, x, w3, b3, out=None):
        B, C, H, W = x.shape
        out = alias_of(out) if out is not None else new_act(B, C, 2 * H, 2 * W, x.dtype, x.device)
        if tuple(out.shape) != (B, C, 2 * H, 2 * W) or out.dtype != x.dtype:
            raise MteError("unpack destination has shape %s, expected %s" % (tuple(out.shape), (B, C, 2 * H, 2 * W)))
        xp, ldx = _pl(x)
        op, ldo = _pl(out)
        w3c, b3c = w3.detach().contiguous().float(), b3.detach().contiguous().float()
        lib.mte_unpack3d_fwd(xp, ldx, w3c.data_ptr(), b3c.data_ptr(), op, ldo, B, H, W, C, _dt(x), _stream())
        ctx.save_for_backward(x, w3c)
        ctx.params = (w3, b3)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w3c = ctx.saved_tensors
        B, C, H, W = x.shape
        dout = as_act(dout, x.dtype)
        dop, ldo = _pl(dout)
        dw3, db3 = _conv3d_weight_grads(lib.mte_unpack3d_bwd_weight, x, dout, *ctx.params)
        dx = new_act(B, C, H, W, x.dtype, x.device)
        dxp, lddx = _pl(dx)
        lib.mte_unpack3d_bwd_data(dop, ldo, w3c.data_ptr(), dxp, lddx, B, H, W, C, _dt(x), _stream())
        return dx, dw3, db3, None


def _rect(src, sy, sx, dst, dy, dx, h, w, mode=0):
    """copy (0) / add (1) / zero (2) a [h x w] pixel rectangle between NHWC activations of equal batch and channels."""
    B, C, Hd, Wd = dst.shape
    dp, ldd = _pl(dst)
    if mode == 2:
        lib.mte_copy_rect(0, 0, 0, 0, 0, 0, dp, ldd, Hd, Wd, dy, dx, B, h, w, C, 2, _dt(dst), _stream())
        return
    sp, lds_ = _pl(src)
    lib.mte_copy_rect(sp, lds_, src.shape[2], src.shape[3], sy, sx, dp, ldd, Hd, Wd, dy, dx, B, h, w, C, mode, _dt(dst), _stream())


class _RectOp(ctypes.Structure):          # mte_rect_op of include/mte_kernels.h
    _fields_ = [("src", ctypes.c_void_p), ("lds_", ctypes.c_long), ("Hs", ctypes.c_int), ("Ws", ctypes.c_int), ("sy", ctypes.c_int),
                ("sx", ctypes.c_int), ("dst", ctypes.c_void_p), ("ldd", ctypes.c_long), ("Hd", ctypes.c_int), ("Wd", ctypes.c_int),
                ("dy", ctypes.c_int), ("dx", ctypes.c_int), ("h", ctypes.c_int), ("w", ctypes.c_int), ("mode", ctypes.c_int)]


def _rects(ops):
    """Several _rect operations in one launch.  ops: (src or None, sy, sx, dst, dy, dx, h, w, mode) with equal batch/channels."""
    arr = (_RectOp * len(ops))()
    B = C = dt = None
    for o, (src, sy, sx, dst, dy, dx, h, w, mode) in zip(arr, ops):
        Bd, Cd, Hd, Wd = dst.shape
        if B is None:
            B, C, dt = Bd, Cd, _dt(dst)
        elif (Bd, Cd, _dt(dst)) != (B, C, dt):
            raise MteError("rectangle batch: mixed batch / channel / dtype")
        dp, ldd = _pl(dst)
        o.dst, o.ldd, o.Hd, o.Wd, o.dy, o.dx, o.h, o.w, o.mode = dp, ldd, Hd, Wd, dy, dx, h, w, mode
        if mode != 2:
            sp, lds_ = _pl(src)
            o.src, o.lds_, o.Hs, o.Ws, o.sy, o.sx = sp, lds_, src.shape[2], src.shape[3], sy, sx
    lib.mte_copy_rects(ctypes.addressof(arr), len(ops), B, C, dt, _stream())


def _deliver(p, t):
    """hand a finished parameter gradient to the sink (flat buffer) or back to autograd"""
    sk = _sink["active"]
    if sk is not None:
        v = sk.claim(p)
        if v is not None:
            v.copy_(t.view(v.shape))
            sk.ready(p)
            return None
    return t


def _deliver_pair(w, b, rec, nw):
    """(dW, db) = (rec[:nw], rec[nw:]) handed to the sink by ONE launch when both parameters have a view there (mte_split_record), else by _deliver"""
    sk = _sink["active"]
    if sk is not None:
        vw, vb = sk.lookup(w), sk.lookup(b)
        if vw is not None and vb is not None and vw.is_contiguous() and vb.is_contiguous() and vw.numel() == nw and vb.numel() == rec.numel() - nw:
            sk.claim(w)
            sk.claim(b)
            lib.mte_split_record(rec.data_ptr(), vw.data_ptr(), nw, vb.data_ptr(), rec.numel() - nw, _stream())
            sk.ready(w)
            sk.ready(b)
            return None, None
    return _deliver(w, rec[:nw].view(w.shape)), _deliver(b, rec[nw:])


def _claim_adjacent(*params):
    """ONE flat fp32 destination over the sink views of `params` when they lie back to back in the flat gradient buffer, else None.  A kernel that produces
    [dW; db] as one record then writes it in place instead of two `_deliver` copies (hipMemcpyAsync = a blit kernel of ~4 us each).  NOTE: FlatParameters' default
    layout (64-element alignment, reverse registration order) does not put a layer's weight and bias back to back, so with the shipped trainer this returns None and
    the records are delivered by copies (profiles/r04_late_steps.txt, 9); a flat buffer built with align=1, reverse=False qualifies.  The caller announces with _announce()."""
    sk = _sink["active"]
    if sk is None:
        return None
    views = [sk.lookup(p) for p in params]
    if any(v is None or not v.is_contiguous() or v.dtype != torch.float32 for v in views):
        return None
    ptr = views[0].data_ptr()
    for v in views:
        if v.data_ptr() != ptr:
            return None
        ptr += 4 * v.numel()
    for p in params:
        sk.claim(p)
    return views[0].as_strided((sum(v.numel() for v in views),), (1,))


def _announce(*params):
    for p in params:
        _sink["active"].ready(p)


def _all_sunk(*params):
    sk = _sink["active"]
    return sk is not None and all(sk.lookup(p) is not None for p in params)


def _conv3d_weight_grads(kernel, x, dout, w3, b3):
    """(dW3, db3) of a Conv3d(1->4) pack/unpack op; queued on the weight-gradient side stream when both land in the sink."""
    B, C, H, W = x.shape

    def run():
        dop, ldo = _pl(dout)
        xp, ldx = _pl(x)
        dst = _claim_adjacent(w3, b3)                       # (the record [dW3 (108); db3 (4)] straight into the flat gradient buffer when the two views are adjacent)
        dwb = dst if dst is not None else torch.empty((112,), dtype=torch.float32, device=x.device)
        kernel(xp, ldx, dop, ldo, dwb.data_ptr(), B, H, W, C, _dt(x), _stream())
        if dst is not None:
            _announce(w3, b3)
            return None, None
        return _deliver_pair(w3, b3, dwb, 108)

    if _side["enabled"] and _all_sunk(w3, b3):
        with torch.cuda.stream(_side_stream_for(x, dout)):
            return run()
    return run()


def pack_fold_applicable(H2, W2, k):
    """Fold only when the exact-border bands are a small part of the image."""
    hb = 2 * (k // 2) + 1
    return H2 > 2 * hb and W2 > 2 * hb and (2.0 * hb / H2 + 2.0 * hb / W2) <= _cfg["pack_fold_max_overhead"]


def _fold_key(w, w3, b, b3):
    return (weights_epoch(), w.data_ptr(), w._version, w3.data_ptr(), w3._version, b.data_ptr(), b._version, b3.data_ptr(), b3._version)


def _fold_now(f, w, w3, b, b3):
    co, C4, k = f["geom"]
    lib.mte_fold_pack_weights(w.detach().data_ptr(), w3.detach().contiguous().float().data_ptr(), b.detach().data_ptr(),
                              b3.detach().contiguous().float().data_ptr(), f["Wf"].data_ptr(), f["bf"].data_ptr(), co, C4, k, _stream())
    f["key"] = _fold_key(w, w3, b, b3)


def _folded_weights(pack_fold, w, w3, b, b3, co, C, k, dev):
    """(W', b') of the folded pack convolution, kept per layer and re-folded only when a parameter changed: by the optimizer hook
    below on the weight-pack side stream right after the update (70 us + two pack kernels per layer that used to sit on the main
    stream in front of the convolution), or here, inline, on first use / after an out-of-band parameter change."""
    f = getattr(pack_fold, "fold", None)
    if f is None or f["geom"] != (co, 4 * C, k) or f["Wf"].device != dev:
        f = pack_fold.fold = {"geom": (co, 4 * C, k), "key": None,
                              "Wf": torch.empty((co, 4 * C, k + 2, k + 2), dtype=torch.float32, device=dev),
                              "bf": torch.empty((co,), dtype=torch.float32, device=dev)}
        _fold_hooks.append(weakref.ref(pack_fold))
    f["params"] = tuple(weakref.ref(t) for t in (w, w3, b, b3))
    if f["key"] != _fold_key(w, w3, b, b3) or _FOLD_INLINE:
        _fold_now(f, w, w3, b, b3)
        pack_fold.key = None                               # the pack of W' is stale too
    return f["Wf"], f["bf"]


_FOLD_INLINE = bool(os.environ.get("MTE_FOLD_INLINE"))     # development A/B: re-fold in every forward pass (the round-1 behaviour)
_fold_hooks = []            # weak references to the fold WeightPacks (prefetch_weight_packs re-folds them before it re-packs)


def _refold_all():
    """called by prefetch_weight_packs on the side stream, after the optimizer changed the parameters"""
    live = []
    for r in _fold_hooks:
        pk = r()
        f = getattr(pk, "fold", None) if pk is not None else None
        if f is None:
            continue
        live.append(r)
        ps = [p() for p in f.get("params", ())]
        if len(ps) == 4 and all(p is not None for p in ps) and f["key"] != _fold_key(*ps):
            _fold_now(f, *ps)
    _fold_hooks[:] = live


_unshuffle_ok = {}           # (B, H2, W2, cout_p, 4C, k) -> the library took mte_conv2d_igemm_unshuffle for this shape


def _dgrad_unshuffled(dy, Wf, pack, dx, k):
    """data gradient of a folded pack convolution written straight into the UN-shuffled dx [B,C,H,W] (mte_conv2d_igemm_unshuffle: no packed gradient tensor, no
    pixel-shuffle pass).  -> False where the library does not take that form for the shape (decided by the first call per shape; the caller then runs the
    two-launch path, bit-identical).  Layers with 4C > 256 keep the 8-phase implicit GEMM + shuffle: there the older tile form would cost more than the pass saves."""
    B, C, H, W = dx.shape
    cp = dy.shape[1]
    if dy.dtype != torch.bfloat16 or 4 * C > 256 or os.environ.get("MTE_NO_UNSHUFFLE") == "1":
        return False
    key = (B, H // 2, W // 2, cp, 4 * C, k)
    ok = _unshuffle_ok.get(key)
    if ok is False:
        return False
    _, wb = pack.get(Wf, dy.dtype, True)
    dyp, lddy = _pl(dy)
    dxp, lddx = _pl(dx)
    args = (dyp, lddy, wb.data_ptr(), dxp, lddx, B, H // 2, W // 2, cp, 4 * C, k, k, _dt(dy), 0, _stream())
    if ok is None:                                           # first call for this shape: the raw entry point, to see MTE_ERR_UNSUPPORTED instead of an exception
        from . import _lib as _L
        rc = _L.lib.load().mte_conv2d_igemm_unshuffle(*args)
        if rc not in (0, -3):                                # (-3 = MTE_ERR_UNSUPPORTED, include/mte_kernels.h)
            raise MteError("mte_conv2d_igemm_unshuffle failed: %s" % rc)
        _unshuffle_ok[key] = rc == 0
        return rc == 0
    lib.mte_conv2d_igemm_unshuffle(*args)
    return True


class PackFoldedConvGnEluFn(torch.autograd.Function):
    """PackLayerConv3d as ONE (k+2)x(k+2) convolution over the packed tensor: conv3d(1->4) is folded into the k x k conv
    weights (csrc/pack_fold.hip), halving the MACs of pack1 and removing the 16C-channel intermediate.  The k/2-pixel
    border, where zero padding sits between the two reference ops (layers01.py:237-238 vs :31), is recomputed exactly
    with the unfolded kernels on four thin bands and pasted over the folded result; backward mirrors it."""

    @staticmethod
    def forward(ctx, x, w3, b3, w, b, gamma, beta, pack_unf, pack_fold, out=None):
        B, C, H, W = x.shape
        H2, W2 = H // 2, W // 2
        co, _, k, _ = w.shape
        pad, hb = k // 2, 2 * (k // 2) + 1
        dt, dev = _dt(x), x.device
        w3c, b3c = w3.detach().contiguous().float(), b3.detach().contiguous().float()
        # exact border bands: group 1 = top/bottom rows, group 2 = left/right columns (rows pad .. H2-pad) -- a chain of small launches,
        # on its own stream beside the interior path
        bands = _BandChain()
        with bands:
            wu, _ = pack_unf.get(w, x.dtype, False)
            xb1 = new_act(2 * B, C, 2 * hb, W, x.dtype, dev)
            xb2 = new_act(2 * B, C, H, 2 * hb, x.dtype, dev)
            _rects([(x, 0, 0, xb1[:B], 0, 0, 2 * hb, W, 0), (x, H - 2 * hb, 0, xb1[B:], 0, 0, 2 * hb, W, 0),
                    (x, 0, 0, xb2[:B], 0, 0, H, 2 * hb, 0), (x, 0, W - 2 * hb, xb2[B:], 0, 0, H, 2 * hb, 0)])
            Tb = []
            for xb, (hh, ww) in ((xb1, (hb, W2)), (xb2, (H2, hb))):
                T = new_act(2 * B, 16 * C, hh, ww, x.dtype, dev)
                sp, lds_ = _pl(xb)
                tp, ldt = _pl(T)
                lib.mte_pack3d_fwd(sp, lds_, w3c.data_ptr(), b3c.data_ptr(), tp, ldt, 2 * B, xb.shape[2], xb.shape[3], C, dt, _stream())
                Tb.append(T)
            yb1 = conv_forward(Tb[0], wu, b, co, k, k, pack=pack_unf, w=w)
            yb2 = conv_forward(Tb[1], wu, b, co, k, k, pack=pack_unf, w=w)
        # folded weights + interior result
        P = new_act(B, 4 * C, H2, W2, x.dtype, dev)
        xp, ldx = _pl(x)
        pp, ldp = _pl(P)
        lib.mte_pixel_shuffle(xp, ldx, pp, ldp, B, H, W, C, 0, dt, _stream())
        Wf, bf = _folded_weights(pack_fold, w, w3, b, b3, co, C, k, dev)
        wfp, _ = pack_fold.get(Wf, x.dtype, False)
        y = conv_forward(P, wfp, bf, co, k + 2, k + 2, pack=pack_fold, w=Wf)
        bands.join()
        _rects([(yb1[:B], 0, 0, y, 0, 0, pad, W2, 0), (yb1[B:], hb - pad, 0, y, H2 - pad, 0, pad, W2, 0),
                (yb2[:B], pad, 0, y, pad, 0, H2 - 2 * pad, pad, 0), (yb2[B:], pad, hb - pad, y, pad, W2 - pad, H2 - 2 * pad, pad, 0)])
        z, stats = _gn_forward(y, None, None, gamma, beta, GN_EPS, out=None if out is None else alias_of(out))
        ctx.save_for_backward(P, xb1, xb2, Tb[0], Tb[1], y, stats, w, w3c, b3c, gamma, beta, Wf)
        ctx.params = (w3, b3, b)
        ctx.packs = (pack_unf, pack_fold)
        ctx.geom = (B, C, H, W, co, k)
        return z

    @staticmethod
    def backward(ctx, dz):
        P, xb1, xb2, T1, T2, y, stats, w, w3c, b3c, gamma, beta, Wf = ctx.saved_tensors
        w3, b3, b = ctx.params
        pack_unf, pack_fold = ctx.packs
        B, C, H, W, co, k = ctx.geom
        H2, W2, pad, hb = H // 2, W // 2, k // 2, 2 * (k // 2) + 1
        dt, dev = _dt(P), P.device
        gg, sg = _grad_dst(gamma, zero=True)
        gbt, sbt = _grad_dst(beta, zero=True)
        dy, _, dgamma, dbeta = _gn_backward(dz, y, None, None, stats, gamma, beta, GN_EPS, False, dgamma=gg, dbeta=gbt)
        # ---- band paths (unfolded, exact), beside the interior path
        bands = _BandChain()
        with bands:
            dyb1 = torch.zeros((2 * B, hb, W2, co), dtype=P.dtype, device=dev).permute(0, 3, 1, 2)
            dyb2 = torch.zeros((2 * B, H2, hb, co), dtype=P.dtype, device=dev).permute(0, 3, 1, 2)
            _rects([(dy, 0, 0, dyb1[:B], 0, 0, pad, W2, 0), (dy, H2 - pad, 0, dyb1[B:], hb - pad, 0, pad, W2, 0),
                    (dy, pad, 0, dyb2[:B], pad, 0, H2 - 2 * pad, pad, 0), (dy, pad, W2 - pad, dyb2[B:], pad, hb - pad, H2 - 2 * pad, pad, 0)])
            gathered = bands.mark()                            # the band pixels of dy have been read (they are cleared below)
            dTs, dxb = [], []
            for xb, T, dyb in ((xb1, T1, dyb1), (xb2, T2, dyb2)):
                dT, _, _ = conv_backward(T, dyb, w, pack_unf, True, need_dw=False)
                Bb, _, hx, wx = xb.shape
                tp, ldt = _pl(dT)
                dxi = new_act(Bb, C, hx, wx, P.dtype, dev)
                dp_, ldd = _pl(dxi)
                lib.mte_pack3d_bwd_data(tp, ldt, w3c.data_ptr(), dp_, ldd, Bb, hx, wx, C, dt, _stream())
                dTs.append(dT)
                dxb.append(dxi)
        # ---- interior path (folded): band pixels carry no gradient here
        bands.join(gathered)
        _rects([(None, 0, 0, dy, 0, 0, pad, W2, 2), (None, 0, 0, dy, H2 - pad, 0, pad, W2, 2),
                (None, 0, 0, dy, pad, 0, H2 - 2 * pad, pad, 2), (None, 0, 0, dy, pad, W2 - pad, H2 - 2 * pad, pad, 2)])

        def weight_grads():
            """every parameter gradient of the layer except gamma/beta; nothing on the data-gradient chain needs them"""
            bands.join()                                       # (on the stream that runs the weight gradients: they read dTs / dyb)
            sw = _stream()
            dw_band = db_band = None
            # (round 4) the parameter gradients are formed IN their places in the flat gradient buffer where the sink hands them out (zero after zero_grad):
            # the four `_deliver` copies of this layer were eight ~4 us blit kernels
            dst3 = _claim_adjacent(w3, b3)
            dk3b = dst3 if dst3 is not None else torch.zeros((112,), dtype=torch.float32, device=dev)
            gw_dst, gw_sunk = _grad_dst(w)
            gb_dst, gb_sunk = _grad_dst(b)
            for xb, T, dyb, dT in ((xb1, T1, dyb1, dTs[0]), (xb2, T2, dyb2, dTs[1])):
                dwi, dbi = _conv_wgrad(T, dyb, w, True, gw_dst if dw_band is None else None, None)
                dw_band = dwi if dw_band is None else dw_band.add_(dwi)
                db_band = dbi if db_band is None else db_band.add_(dbi)
                Bb, _, hx, wx = xb.shape
                tmp = torch.empty((112,), dtype=torch.float32, device=dev)
                sp, lds_ = _pl(xb)
                tp, ldt = _pl(dT)
                lib.mte_pack3d_bwd_weight(sp, lds_, tp, ldt, tmp.data_ptr(), Bb, hx, wx, C, dt, sw)
                dk3b.add_(tmp)
            dWf, dbf = _conv_wgrad(P, dy, Wf, True, None, None)
            lib.mte_unfold_pack_wgrad(dWf.data_ptr(), dbf.data_ptr(), w.detach().data_ptr(), w3c.data_ptr(), b3c.data_ptr(),
                                      dw_band.data_ptr(), dk3b.data_ptr(), co, 4 * C, k, 1, sw)
            db = torch.add(dbf, db_band, out=gb_dst)
            if dst3 is not None:
                _announce(w3, b3)
                g3 = gb3 = None
            else:
                g3, gb3 = _deliver_pair(w3, b3, dk3b, 108)
            return (g3, gb3, _grad_ret(w, dw_band, gw_sunk), _grad_ret(b, db, gb_sunk))

        if _side["enabled"] and _all_sunk(w3, b3, w, b):
            with torch.cuda.stream(_side_stream_for(P, dy, T1, T2, dyb1, dyb2, dTs[0], dTs[1], xb1, xb2, Wf)):
                g3, gb3, gw, gb = weight_grads()
        else:
            g3, gb3, gw, gb = weight_grads()
        dx = new_act(B, C, H, W, P.dtype, dev)
        if not _dgrad_unshuffled(dy, Wf, pack_fold, dx, k + 2):
            dP, _, _ = conv_backward(P, dy, Wf, pack_fold, True, need_dw=False)
            sp, lds_ = _pl(dP)
            dp_, ldd = _pl(dx)
            lib.mte_pixel_shuffle(sp, lds_, dp_, ldd, B, H, W, C, 1, dt, _stream())
        bands.join()
        # (the four bands overlap in the corners: two launches so that no element is read-modified by two operations at once)
        _rects([(dxb[0][:B], 0, 0, dx, 0, 0, 2 * hb, W, 1), (dxb[0][B:], 0, 0, dx, H - 2 * hb, 0, 2 * hb, W, 1)])
        _rects([(dxb[1][:B], 0, 0, dx, 0, 0, H, 2 * hb, 1), (dxb[1][B:], 0, 0, dx, 0, W - 2 * hb, H, 2 * hb, 1)])
        return (dx, g3, gb3, gw, gb, _grad_ret(gamma, dgamma, sg), _grad_ret(beta, dbeta, sbt), None, None, None)


class InvDepthFn(torch.autograd.Function):
    """sigmoid(conv3x3(pad1(x)) + b) / min_depth -> fp32 [B,1,H,W]  (InvDepth.forward, layers01.py:120-123)."""

    @staticmethod
    def forward(ctx, x, w, b, min_depth):
        B, C, H, W = x.shape
        out = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
        xp, ldx = _pl(x)
        wc = w.detach().contiguous().float()
        lib.mte_invdepth_fwd(xp, ldx, wc.data_ptr(), b.detach().float().data_ptr(), out.data_ptr(), B, H, W, C, min_depth, _dt(x), _stream())
        ctx.save_for_backward(x, wc, out)
        ctx.min_depth = min_depth
        ctx.params = (w, b)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wc, out = ctx.saved_tensors
        w, b = ctx.params
        B, C, H, W = x.shape
        dout = dout.contiguous().float()
        dx = new_act(B, C, H, W, x.dtype, x.device)
        dlogit = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
        dxp, lddx = _pl(dx)
        lib.mte_invdepth_bwd_data(wc.data_ptr(), out.data_ptr(), dout.data_ptr(), dlogit.data_ptr(), dxp, lddx,
                                  B, H, W, C, ctx.min_depth, _dt(x), _stream())

        def weight_grads():
            xp, ldx = _pl(x)
            dst = _claim_adjacent(w, b) if tuple(w.shape) == (1, C, 3, 3) and b.numel() == 1 else None
            dwb = dst if dst is not None else torch.empty((C * 9 + 1,), dtype=torch.float32, device=x.device)
            rec = torch.empty((int(lib.mte_invdepth_bwd_weight_workspace_elems(C)),), dtype=torch.float32, device=x.device)
            lib.mte_invdepth_bwd_weight(xp, ldx, dlogit.data_ptr(), dwb.data_ptr(), rec.data_ptr(), B, H, W, C, _dt(x), _stream())
            if dst is not None:
                _announce(w, b)
                return None, None
            return _deliver_pair(w, b, dwb, C * 9)

        if _side["enabled"] and _all_sunk(w, b):
            with torch.cuda.stream(_side_stream_for(x, dlogit)):
                dw, db = weight_grads()
        else:
            dw, db = weight_grads()
        return dx, dw, db, None


def _pending_slot(slot):
    """the slot (this one or an ancestor) whose buffer is still waiting for a deferred 1x1 data gradient (conv_backward), or None"""
    while slot is not None:
        if slot["buf"] is not None:
            return slot if "pending" in slot else None
        slot = slot["parent"]
    return None


def _flush_pending(slot):
    """launch the deferred 1x1 data gradient of `slot` on its own (plain store into the slot's buffer): whoever needs the buffer's contents before the
    3x3 consumer that would have carried it came along"""
    dy3, w3, pack3 = slot.pop("pending")
    dx = slot["buf"]
    B, Cp, H, W = dx.shape
    cout = w3.shape[0]
    d3p, ldd3 = _pl(dy3)
    dxp, lddx = _pl(dx)
    lib.mte_conv2d_patch_fwd(d3p, ldd3, pack3.get_patch(w3, 'b').data_ptr(), 0, dxp, lddx, B, H, W, cout, Cp, 1, 1, 0, _stream())


def _fork_target(slot, shape, dtype, keep_pending=False):
    """The buffer a data gradient should be ACCUMULATED into: the gradient that another consumer of the same forked
    activation (or, for a fork of a fork, of its parent) has already produced.  None = this consumer is the first.
    keep_pending: the caller deals with a deferred 1x1 term of that buffer itself (conv_backward); everybody else gets it flushed first."""
    if _cfg.get("no_fork_accumulate"):
        return None
    while slot is not None:
        buf = slot["buf"]
        if buf is not None:
            if "pending" in slot and not keep_pending:
                _flush_pending(slot)
            return buf if (tuple(buf.shape) == tuple(shape) and buf.dtype == dtype and is_act(buf)) else None
        slot = slot["parent"]
    return None


class ForkFn(torch.autograd.Function):
    """An activation with two consumers (residual shortcut + main branch, encoder skip + next stage, decoder feature +
    inv-depth head).  Forward hands out two aliases that carry a shared `slot`; a consumer whose backward produces the data
    gradient through a conv epilogue deposits it there, and the second one accumulates into the same buffer
    (`accumulate` of mte_conv2d_igemm / mte_conv2d_patch_fwd; a decoder concat deposits its channel-slice view), so the
    autograd sum of the two gradients usually costs no pass at all.  Otherwise backward adds them in one strided-view-aware
    pass (autograd's generic add is 3x slower on channel-slice gradients)."""

    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        return alias_of(x), alias_of(x)

    @staticmethod
    def backward(ctx, g1, g2):
        slot = ctx.slot
        if "pending" in slot:                                # nobody carried the deferred 1x1 term: its buffer is one of g1 / g2
            _flush_pending(slot)
        if g1 is None or g2 is None:
            out = g1 if g2 is None else g2
        elif g1.data_ptr() == g2.data_ptr() and g1.shape == g2.shape and g1.stride() == g2.stride():
            out = g1                                         # the consumers shared one buffer: it already holds the sum
        else:
            dt = g1.dtype if g1.dtype == g2.dtype else compute_dtype()
            g1, g2 = as_act(g1, dt), as_act(g2, dt)
            B, C, H, W = g1.shape
            out = new_act(B, C, H, W, dt, g1.device)
            p1, l1 = _pl(g1)
            p2, l2 = _pl(g2)
            po, lo = _pl(out)
            lib.mte_add_channels(p1, l1, p2, l2, po, lo, B * H * W, C, _dt(out), _stream())
        slot["buf"] = None
        return out, None


def fork(x):
    """-> two aliases of the NHWC activation x for its two consumers (see ForkFn); no-op without gradients"""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x
    slot = {"buf": None, "parent": getattr(x, "_mte_fork_slot", None)}
    a, b = ForkFn.apply(x, slot)
    a._mte_fork_slot = slot
    b._mte_fork_slot = slot
    return a, b


class ConcatFn(torch.autograd.Function):
    """torch.cat((parts...[, nearest_up2(inv)]), 1) into one NHWC buffer whose channel count is padded to a
    multiple of 8 (decoder version 'A', PackNetSAN01.py:105-143).  Backward hands out channel-slice views."""

    @staticmethod
    def forward(ctx, inv, buf, *parts):
        """`buf` (optional): pre-allocated concat buffer (new_concat_buffer); parts that already live in their channel
        block of it (producers were given channel_slice(buf, ..) as destination) are not copied."""
        B, _, H, W = parts[0].shape
        dtype = parts[0].dtype
        chans = [p.shape[1] for p in parts]
        ctot = sum(chans) + (8 if inv is not None else 0)
        if buf is None:
            buf = new_act(B, ctot, H, W, dtype, parts[0].device)
        else:
            if tuple(buf.shape) != (B, ctot, H, W) or buf.dtype != dtype:
                raise MteError("concat buffer has shape %s, expected %s" % (tuple(buf.shape), (B, ctot, H, W)))
            buf = alias_of(buf)
        st = _stream()
        off = 0
        for p, c in zip(parts, chans):
            dst = buf[:, off:off + c]
            off += c
            if p.dtype == dtype and p.data_ptr() == dst.data_ptr() and p.stride() == dst.stride():
                continue                                     # produced in place
            p = as_act(p, dtype)
            sp, lds_ = _pl(p)
            dp, ldd = _pl(dst)
            lib.mte_copy_channels(sp, lds_, dp, ldd, B * H * W, c, _dt(p), st)
        if inv is not None:
            dst = buf[:, off:off + 8]
            dp, ldd = _pl(dst)
            lib.mte_upsample_inv_fwd(inv.contiguous().data_ptr(), dp, ldd, B, H // 2, W // 2, _dt(buf), st)
        ctx.chans = chans
        ctx.has_inv = inv is not None
        ctx.slots = [getattr(p, "_mte_fork_slot", None) for p in parts]
        return buf

    @staticmethod
    def backward(ctx, dbuf):
        dbuf = as_act(dbuf)
        B, _, H, W = dbuf.shape
        outs = []
        off = 0
        for c, slot in zip(ctx.chans, ctx.slots):
            g = dbuf[:, off:off + c]
            if slot is not None and slot["buf"] is None and _fork_target(slot["parent"], g.shape, g.dtype) is None:
                slot["buf"] = g                              # the other consumer's conv epilogue will add into this slice
            outs.append(g)
            off += c
        dinv = None
        if ctx.has_inv:
            dinv = torch.empty((B, 1, H // 2, W // 2), dtype=torch.float32, device=dbuf.device)
            src = dbuf[:, off:off + 8]
            sp, lds_ = _pl(src)
            lib.mte_upsample_inv_bwd(sp, lds_, dinv.data_ptr(), B, H // 2, W // 2, 0, _dt(dbuf), _stream())
        return (dinv, None) + tuple(outs)


def new_concat_buffer(B, chans, with_inv, H, W, dtype=None, device="cuda"):
    """Decoder concat buffer for ConcatFn: channel blocks `chans` (+ one 8-channel block for an up-sampled inv-depth map)."""
    c = sum(chans) + (8 if with_inv else 0)
    ld = (c + 31) // 32 * 32
    if ld == c or os.environ.get("MTE_CONCAT_TIGHT"):
        return new_act(B, c, H, W, dtype or compute_dtype(), device)
    # pixel stride padded to a multiple of 64 bytes (72 -> 96, 136 -> 160, 200 -> 224 channels): with 144 / 272 / 400 bytes per pixel
    # every channel block of every pixel starts inside a 64-byte sector and the producers' 64-byte runs straddle two of them --
    # the GroupNorm pass writing the stem's skip into iconv1's buffer ran at 2.0 TB/s against 5.8 TB/s for the same pass into a
    # dense tensor.  The logical channel count is unchanged (kernels take pointer + pixel stride), the tail is never touched.
    return alias_of(new_act(B, ld, H, W, dtype or compute_dtype(), device)[:, :c])


def image_to_act(rgb, flip=False, dtype=None):
    """fp32 NCHW image -> NHWC compute-dtype activation, channels zero-padded to a multiple of 8."""
    _require_gpu(rgb)
    B, C, H, W = rgb.shape
    rgb = rgb.contiguous().float()
    out = new_act(B, round8(C), H, W, dtype or compute_dtype(), rgb.device)
    op, ldo = _pl(out)
    lib.mte_nchw_to_nhwc(rgb.data_ptr(), op, ldo, B, C, H, W, round8(C), 1 if flip else 0, _dt(out), _stream())
    return out


# --------------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------------


# ---- the loss-side / optimizer-side bindings live in kernels_loss.py (round 6); re-exported so that `kernels.<name>` keeps working
from .kernels_loss import EdgeLossFn, BilinearResizeFn, DepthLossesFn, SilogFn, _EdgeScale, adam_step_flat  # noqa: E402,F401
