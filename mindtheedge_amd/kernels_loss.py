"""Loss-side and optimizer-side bindings of the C ABI (split out of kernels.py in round 6: the fused depth-edge + silog losses, the bilinear resize of the
edge loss and the host form of the Adam step).  Reference: packnet_sfm/losses/grad_loss.py:15-177 (GradLayer / GradLoss / comp_cross_entropy),
losses/supervised_loss.py (silog), models/SemiSupEdgeModel.py (all-scales edge loss), models/model_wrapper.py:142-180 (Adam).  `kernels` re-exports every
name defined here, so callers keep writing kernels.DepthLossesFn etc."""
import ctypes

import torch

from . import kernels as _K          # (imported at the END of kernels.py: its helpers are looked up at call time)
from ._lib import MteError          # noqa: F401  (the library handle is kernels.lib at call time: bench.py swaps it for a timing proxy)


class EdgeLossFn(torch.autograd.Function):
    """weight * class-balanced BCE of sigmoid(directional Sobel(depth) - thresh) against soft edge labels.
    GradLoss.forward ('cross_entropy'), grad_loss.py:122-219, fused with inv2depth when from_inv."""

    @staticmethod
    def forward(ctx, pred, edge, normal, mask, weight, pos_to_neg, from_inv, is_grad, is_sigmoid, thresh, want_gmap):
        B, _, H, W = edge.shape
        pred, edge = pred.contiguous().float(), edge.contiguous().float()
        normal = None if normal is None else normal.contiguous().float()
        mask = None if mask is None else mask.contiguous().float()
        dev = pred.device
        sums = _K._zeros((_K.lib.mte_edge_loss_sums_elems(B, H, W),), torch.float64, dev)
        coef = torch.empty((2 * B + 1,), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        gmap = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev) if want_gmap else None
        st = _K._stream()
        _K.lib.mte_edge_loss_fwd(pred.data_ptr(), edge.data_ptr(), _K._ptr(normal), _K._ptr(mask), sums.data_ptr(), _K._ptr(gmap),
                              B, H, W, int(from_inv), int(is_grad), int(is_sigmoid), float(thresh), st)
        _K.lib.mte_edge_loss_finalize(sums.data_ptr(), B, edge.numel(), float(weight), float(pos_to_neg), int(mask is not None),
                                   1.0, 0, loss.data_ptr(), coef.data_ptr(), st)
        ctx.save_for_backward(pred, edge, normal, mask, coef)
        ctx.cfg = (B, H, W, int(from_inv), int(is_grad), int(is_sigmoid), float(thresh))
        if want_gmap:
            ctx.mark_non_differentiable(gmap)
        return loss, gmap

    @staticmethod
    def backward(ctx, gloss, _g):
        pred, edge, normal, mask, coef = ctx.saved_tensors
        B, H, W, from_inv, is_grad, is_sigmoid, thresh = ctx.cfg
        dpred = torch.empty_like(pred)
        gl = gloss.contiguous().float()
        _K.lib.mte_edge_loss_bwd(pred.data_ptr(), edge.data_ptr(), _K._ptr(normal), _K._ptr(mask), coef.data_ptr(), gl.data_ptr(),
                              dpred.data_ptr(), B, H, W, from_inv, is_grad, is_sigmoid, thresh, _K._stream())
        return (dpred,) + (None,) * 10


class BilinearResizeFn(torch.autograd.Function):
    """F.interpolate(x, size=(H, W), mode='bilinear') for fp32 [B,1,h,w] maps -- the resize GradLoss.forward applies when the
    prediction and the label differ in size (grad_loss.py:127)."""

    @staticmethod
    def forward(ctx, x, H, W):
        x = x.contiguous().float()
        B, C, h, w = x.shape
        y = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        _K.lib.mte_resize_bilinear_fwd(x.data_ptr(), y.data_ptr(), B * C, h, w, H, W, _K._stream())
        ctx.geom = (B, C, h, w, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, h, w, H, W = ctx.geom
        dy = dy.contiguous().float()
        dx = torch.empty((B, C, h, w), dtype=torch.float32, device=dy.device)
        _K.lib.mte_resize_bilinear_bwd(dy.data_ptr(), dx.data_ptr(), B * C, h, w, H, W, _K._stream())
        return dx, None, None


class _EdgeScale(ctypes.Structure):       # mte_edge_scale of include/mte_kernels.h
    _fields_ = [("pred", ctypes.c_void_p), ("edge", ctypes.c_void_p), ("normal", ctypes.c_void_p), ("mask", ctypes.c_void_p),
                ("gmap", ctypes.c_void_p), ("dpred", ctypes.c_void_p), ("H", ctypes.c_int), ("W", ctypes.c_int)]


class DepthLossesFn(torch.autograd.Function):
    """Every loss term of SemiSupEdgeModel.forward (models/SemiSupEdgeModel.py:137-151) in ONE forward and ONE backward launch:
    the depth-edge loss of all scales (compute_edge_loss_with_all_scales, :164-198; GradLoss 'cross_entropy' fused with
    inv2depth) and, when `gt_depth` is given, the sparse silog loss of scale 0 -- both read the same inverse-depth maps.
    -> fp32 [S] (+1): weight * balanced BCE per scale, then the silog loss when gt_depth is given."""

    @staticmethod
    def forward(ctx, weight, pos_to_neg, thresh, from_inv, mask, gt_depth, edges, normals, *preds):
        S = len(preds)
        B = preds[0].shape[0]
        dev = preds[0].device
        preds = [p.contiguous().float() for p in preds]
        edges = [e.contiguous().float() for e in edges]
        normals = [None if n is None else n.contiguous().float() for n in normals]
        mask = None if mask is None else mask.contiguous().float()
        if mask is not None and any(tuple(mask.shape[-2:]) != tuple(p.shape[-2:]) for p in preds):
            raise MteError("one full-resolution mask for every scale is an upstream bug that only works with mask=None")
        gt = None if gt_depth is None else gt_depth.contiguous().float()
        arr = (_EdgeScale * S)()
        for o, p, e, n in zip(arr, preds, edges, normals):
            if tuple(p.shape) != tuple(e.shape) or (n is not None and tuple(n.shape) != tuple(e.shape)):
                raise MteError("prediction / label shapes differ: %s vs %s" % (tuple(p.shape), tuple(e.shape)))
            o.pred, o.edge, o.normal, o.mask = p.data_ptr(), e.data_ptr(), _K._ptr(n), _K._ptr(mask)
            o.gmap = o.dpred = None
            o.H, o.W = p.shape[-2], p.shape[-1]
        work = _K._zeros((_K.lib.mte_edge_loss_work_elems(ctypes.addressof(arr), S, B),), torch.float64, dev)
        losses = torch.empty((S + (1 if gt is not None else 0),), dtype=torch.float32, device=dev)
        coef = torch.empty((S * (2 * B + 1),), dtype=torch.float32, device=dev)
        aux = torch.empty((2,), dtype=torch.float32, device=dev) if gt is not None else None
        _K.lib.mte_edge_loss_multi_fwd(ctypes.addressof(arr), S, B, int(from_inv), 1, 1, float(thresh), float(weight), float(pos_to_neg),
                                    _K._ptr(gt), work.data_ptr(), losses.data_ptr(), coef.data_ptr(),
                                    losses.data_ptr() + 4 * S if gt is not None else 0, _K._ptr(aux), _K._stream())
        ctx.save_for_backward(coef, mask, gt, aux, *preds, *edges, *[n for n in normals if n is not None])
        ctx.cfg = (S, B, int(from_inv), float(thresh), [n is not None for n in normals])
        return losses

    @staticmethod
    def backward(ctx, glosses):
        S, B, from_inv, thresh, has_n = ctx.cfg
        coef, mask, gt, aux = ctx.saved_tensors[:4]
        rest = ctx.saved_tensors[4:]
        preds, edges, nrm = rest[:S], rest[S:2 * S], list(rest[2 * S:])
        normals = [nrm.pop(0) if h else None for h in has_n]
        dpreds = [torch.empty_like(p) for p in preds]
        arr = (_EdgeScale * S)()
        for o, p, e, n, d in zip(arr, preds, edges, normals, dpreds):
            o.pred, o.edge, o.normal, o.mask, o.gmap, o.dpred = p.data_ptr(), e.data_ptr(), _K._ptr(n), _K._ptr(mask), None, d.data_ptr()
            o.H, o.W = p.shape[-2], p.shape[-1]
        gl = glosses.contiguous().float()
        _K.lib.mte_edge_loss_multi_bwd(ctypes.addressof(arr), S, B, from_inv, 1, 1, thresh, coef.data_ptr(), gl.data_ptr(), _K._ptr(gt),
                                    _K._ptr(aux), gl.data_ptr() + 4 * S if gt is not None else 0, _K._stream())
        return (None,) * 8 + tuple(dpreds)


class SilogFn(torch.autograd.Function):
    """10*sqrt(mean(d^2) - 0.85*mean(d)^2), d = log(10(inv+1e-5)) - log(10/depth) over depth > 0
    (SupervisedLoss 'sparse-silog' at scale 0: supervised_loss.py:57-69,155-216 + depth2inv)."""

    @staticmethod
    def forward(ctx, inv, depth):
        inv, depth = inv.contiguous().float(), depth.contiguous().float()
        dev = inv.device
        sums = torch.empty((3,), dtype=torch.float64, device=dev)
        aux = torch.empty((2,), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _K.lib.mte_silog_fwd(inv.data_ptr(), depth.data_ptr(), inv.numel(), sums.data_ptr(), 1.0, 0, loss.data_ptr(), aux.data_ptr(), _K._stream())
        ctx.save_for_backward(inv, depth, aux)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        inv, depth, aux = ctx.saved_tensors
        dinv = torch.empty_like(inv)
        gl = gloss.contiguous().float()
        _K.lib.mte_silog_bwd(inv.data_ptr(), depth.data_ptr(), aux.data_ptr(), gl.data_ptr(), dinv.data_ptr(), inv.numel(), 0, _K._stream())
        return dinv, None


def adam_step_flat(p, g, m, v, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, gscale=1.0):
    """In-place fused Adam over flat fp32 device buffers."""
    _K._require_gpu(p)
    _K.lib.mte_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(betas[0]), float(betas[1]),
                      float(eps), int(step), float(gscale), _K._stream())
    _K.bump_weights_epoch()
