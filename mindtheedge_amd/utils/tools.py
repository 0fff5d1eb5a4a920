"""Depth-edge annotation post-processing on the device -- same names as the reference's
packnet_sfm/utils/tools.py (``non_max_suppression`` :9-46, ``hysteresis`` :49-92) plus the normal-map quantisation of
infer_edge_estimation.py:194-200 and the per-scale chain of :186-206 (``annotate_edges``).

Inputs are CUDA tensors [H,W] or [B,H,W] (float32); outputs are CUDA tensors of the same leading shape.  There is no
CPU path: a host tensor raises MteError.  ``hysteresis`` is the only function that looks at the device from the host:
one 4-byte flag per ``SWEEPS`` propagation sweeps, to know when the fixed point has been reached.
"""
import torch

SWEEPS = 8


def _maps(img):
    from .. import kernels as K
    K._require_gpu(img)
    if img.dim() not in (2, 3):
        raise ValueError("expected an [H,W] or [B,H,W] map, got {}".format(tuple(img.shape)))
    x = img.detach().float().contiguous()
    return (x.unsqueeze(0) if img.dim() == 2 else x), img.dim() == 2


def _sobel_nms(img, scale, want_normals, want_nms):
    from .. import kernels as K
    x, squeeze = _maps(img)
    B, H, W = x.shape
    normals = torch.empty((B, H, W), dtype=torch.uint8, device=x.device) if want_normals else None
    nms = torch.empty_like(x) if want_nms else None
    K.lib.mte_dee_sobel_nms(x.data_ptr(), float(scale), normals.data_ptr() if want_normals else None,
                            nms.data_ptr() if want_nms else None, B, H, W, K._stream())
    if squeeze:
        normals = normals[0] if want_normals else None
        nms = nms[0] if want_nms else None
    return normals, nms


def non_max_suppression(img):
    """Reference tools.py:9-46: 5x5 Sobel angle quantised to 4 directions, keep local maxima, zero frame."""
    return _sobel_nms(img, 1.0, False, True)[1]


def sobel_normals(img):
    """uint8 normal map of infer_edge_estimation.py:194-200: ((atan2(-sobely, sobelx) in degrees + 180) / 360 * 255)."""
    return _sobel_nms(img, 1.0, True, False)[0]


def hysteresis(img, t_low=0.3, t_high=0.7):
    """Reference tools.py:49-92 (including what it does to the one-pixel frame and the NaN map when no pixel is strong)."""
    from .. import kernels as K
    x, squeeze = _maps(img)
    B, H, W = x.shape
    state = torch.empty((B, H, W), dtype=torch.uint8, device=x.device)
    info = torch.empty(B * 4, dtype=torch.int32, device=x.device)
    flags = torch.empty(SWEEPS + 1, dtype=torch.int32, device=x.device)
    out = torch.empty_like(x)
    K.lib.mte_hysteresis_begin(x.data_ptr(), state.data_ptr(), info.data_ptr(), B, H, W, float(t_low), float(t_high), K._stream())
    for _ in range(H * W // SWEEPS + 2):            # every sweep that is not the last turns at least one pixel: a hard bound
        K.lib.mte_hysteresis_propagate(state.data_ptr(), flags.data_ptr(), SWEEPS, B, H, W, K._stream())
        if int(flags[SWEEPS].item()) == 0:          # the only host read: did the last sweep of this batch still change pixels?
            break
    else:
        raise K.MteError("hysteresis did not reach a fixed point")
    K.lib.mte_hysteresis_finish(x.data_ptr(), state.data_ptr(), info.data_ptr(), out.data_ptr(), B, H, W, K._stream())
    return out[0] if squeeze else out


def annotate_edges(pred_inv_depths, nms=True, hysteresis_=True, normals=True, scales=None):
    """The per-scale chain of infer_edge_estimation.py:186-206 on the network's output list: for every scale
    probability = pred / 2 -> (normals uint8) -> (NMS) -> (hysteresis).  Returns [(edges [B,H,W] float32, normals
    [B,H,W] uint8 or None), ...], one entry per scale; the caller writes edges*255 / normals as PNGs."""
    out = []
    for s, pred in enumerate(pred_inv_depths):
        if scales is not None and s >= scales:
            break
        p = pred[:, 0, :, :] if pred.dim() == 4 else pred
        n, e = _sobel_nms(p, 0.5, normals, nms) if (normals or nms) else (None, None)
        if not nms:
            from .. import kernels as K
            K._require_gpu(p)
            e = p.detach().float() * 0.5
        if hysteresis_:
            e = hysteresis(e)
        out.append((e, n))
    return out
