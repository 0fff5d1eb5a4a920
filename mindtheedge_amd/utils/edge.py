"""Chamfer edge metrics on the device (edge half) -- same names as the reference's
packnet_sfm/utils/edge.py (``chamfer_distance`` :19-64) plus the precision / recall / F1 triple that
``ModelWrapper.compute_edge_metrics`` builds from it (models/model_wrapper.py:426-440).

Inputs are CUDA tensors [H,W] or [B,H,W] on the 0..255 scale (uint8 or float: an edge is v/255 > 0.5, as upstream).
The predicted edge image itself comes from cv2.Canny upstream (model_wrapper.py:396-400); that step is not part of this
build (OpenCV arithmetic, parity unpinned) -- pass any edge image.  No CPU path: host tensors raise MteError.
"""
import torch


def _edge_maps(*images):
    from .. import kernels as K
    out, squeeze = [], False
    for im in images:
        K._require_gpu(im)
        if im.dim() not in (2, 3):
            raise ValueError("expected an [H,W] or [B,H,W] edge image, got {}".format(tuple(im.shape)))
        squeeze = im.dim() == 2
        x = im.detach().float().contiguous()
        out.append(x.unsqueeze(0) if squeeze else x)
    if any(o.shape != out[0].shape for o in out):
        raise ValueError("shape mismatch: {}".format([tuple(o.shape) for o in out]))
    return out, squeeze


def chamfer_distance(im_pred, im_gt, mask=None, edge_to_edge_thresh=5, return_map=True):
    """-> (c_dist, percentage, cond_map) like the reference: float64 device scalars ([B] for a batch) and the -1/0/1 map
    (None with return_map=False)."""
    from .. import kernels as K
    if mask is not None:
        raise NotImplementedError("mask is only used by the reference's colour-image callers; not on this path")
    (p, g), squeeze = _edge_maps(im_pred, im_gt)
    B, H, W = p.shape
    nbytes = K.lib.mte_chamfer_workspace_bytes(B, H, W)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=p.device)
    out = torch.empty((B, 2), dtype=torch.float64, device=p.device)
    cond = torch.empty_like(p) if return_map else None
    K.lib.mte_chamfer_distance(p.data_ptr(), g.data_ptr(), B, H, W, float(edge_to_edge_thresh), ws.data_ptr(), nbytes, out.data_ptr(),
                               None, cond.data_ptr() if return_map else None, K._stream())
    c, perc = out[:, 0], out[:, 1]
    if squeeze:
        return c[0], perc[0], (cond[0] if return_map else None)
    return c, perc, cond


def distance_transform_edt(im_gt):
    """Exact Euclidean distance (float32 map) to the nearest edge pixel of ``im_gt`` -- scipy.ndimage.distance_transform_edt
    of the complement, as used inside chamfer_distance (edge.py:36)."""
    from .. import kernels as K
    (g,), squeeze = _edge_maps(im_gt)
    B, H, W = g.shape
    nbytes = K.lib.mte_chamfer_workspace_bytes(B, H, W)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=g.device)
    out = torch.empty((B, 2), dtype=torch.float64, device=g.device)
    dist = torch.empty_like(g)
    K.lib.mte_chamfer_distance(g.data_ptr(), g.data_ptr(), B, H, W, 5.0, ws.data_ptr(), nbytes, out.data_ptr(), dist.data_ptr(), None, K._stream())
    return dist[0] if squeeze else dist


def edge_precision_recall_f1(im_pred, im_gt, edge_to_edge_thresh=5):
    """(precision, recall, F1) of a predicted edge image against the ground truth, model_wrapper.py:431-438:
    precision = share of predicted pixels near a true edge, recall = share of true pixels near a predicted edge."""
    _, p, _ = chamfer_distance(im_pred, im_gt, edge_to_edge_thresh=edge_to_edge_thresh, return_map=False)
    _, r, _ = chamfer_distance(im_gt, im_pred, edge_to_edge_thresh=edge_to_edge_thresh, return_map=False)
    return p, r, 2 * ((p * r) / (p + r))


def compute_edge_metrics(edge_images, gt_edge, edge_to_edge_thresh=5):
    """The loop of ModelWrapper.compute_edge_metrics (model_wrapper.py:426-440) over already extracted edge images
    (upstream: three cv2.Canny settings, or three thresholds of an edge-probability map): -> float64 device tensor
    [len(edge_images) * 3] of (precision, recall, F1) per image, in the reference's order."""
    out = []
    for im in edge_images:
        out.extend(edge_precision_recall_f1(im, gt_edge, edge_to_edge_thresh))
    return torch.stack([o.reshape(()) for o in out])


CANNY_THRESHOLDS = ((10, 20), (20, 40), (30, 60))       # models/model_wrapper.py:398-400


def canny_from_depth(depth, thresholds=CANNY_THRESHOLDS, return_vis=False):
    """Edge images of a predicted depth map as ModelWrapper.compute_edge_metrics extracts them (model_wrapper.py:396-400):
    ``vis = uint8(depth * (255 / depth.max()))`` per image, then ``cv2.Canny(vis, lo, hi)`` for every threshold pair.
    depth: CUDA [H,W] or [B,H,W] -> float32 [P,B,H,W] (or [P,H,W]) with 255 on edges.

    PARITY UNPINNED: this is OpenCV's published Canny (apertureSize 3, L1 gradient) restated without OpenCV at hand
    (oracle/canny_oracle.py); it is tested against that restatement and known answers only."""
    import ctypes
    from .. import kernels as K
    (d,), squeeze = _edge_maps(depth)
    B, H, W = d.shape
    P = len(thresholds)
    if not 1 <= P <= 4:
        raise ValueError("1..4 threshold pairs")
    th = (ctypes.c_int * (2 * P))(*[int(v) for pair in thresholds for v in pair])
    max_ws = torch.empty(B, dtype=torch.int32, device=d.device)
    vis = torch.empty((B, H, W), dtype=torch.uint8, device=d.device) if return_vis else None
    state = torch.empty((P, B, H, W), dtype=torch.uint8, device=d.device)
    sweeps = 8
    flags = torch.empty(sweeps + 1, dtype=torch.int32, device=d.device)
    K.lib.mte_canny_begin(d.data_ptr(), B, H, W, P, ctypes.addressof(th), max_ws.data_ptr(), vis.data_ptr() if return_vis else None,
                          state.data_ptr(), K._stream())
    for _ in range(H * W // sweeps + 2):            # every sweep that is not the last turns at least one pixel: a hard bound
        K.lib.mte_canny_propagate(state.data_ptr(), flags.data_ptr(), sweeps, P * B, H, W, K._stream())
        if int(flags[sweeps].item()) == 0:              # one host read per 8 sweeps, as in utils/tools.py::hysteresis
            break
    else:
        raise K.MteError("Canny hysteresis did not reach a fixed point")
    edges = torch.empty((P, B, H, W), dtype=torch.float32, device=d.device)
    K.lib.mte_canny_finish(state.data_ptr(), edges.data_ptr(), P * B, H, W, K._stream())
    if squeeze:
        edges, vis = edges[:, 0], (vis[0] if return_vis else None)
    return (edges, vis) if return_vis else edges


def resize_linear(img, shape):
    """cv2.resize(img, (shape[1], shape[0]), interpolation=cv2.INTER_LINEAR) for float maps [H,W] / [B,H,W] on the device
    (model_wrapper.py:386-387).  Parity-unpinned restatement of OpenCV's resize (oracle/canny_oracle.py::resize_linear)."""
    from .. import kernels as K
    (x,), squeeze = _edge_maps(img)
    B, h, w = x.shape
    H, W = int(shape[0]), int(shape[1])
    out = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
    K.lib.mte_resize_linear(x.data_ptr(), B, h, w, out.data_ptr(), H, W, K._stream())
    return out[0] if squeeze else out


def compute_edge_metrics_from_depth(depth, gt_edge, edge_to_edge_thresh=5):
    """ModelWrapper.compute_edge_metrics for depth models (model_wrapper.py:376-440) on ONE [H,W] depth map and its
    ground-truth edge image (0..255 scale): three Canny settings x (precision, recall, F1) -> float64 device tensor [9].
    The depth is first resized to the edge image's size like the reference does (cv2.resize, INTER_LINEAR).  The resize
    and the Canny step are parity-unpinned restatements of OpenCV; the chamfer part is pinned."""
    if depth.dim() != 2 or gt_edge.dim() != 2:
        raise ValueError("one [H,W] depth map and one [H,W] edge image, as the reference evaluates them")
    if depth.shape != gt_edge.shape:
        depth = resize_linear(depth, gt_edge.shape)
    edges = canny_from_depth(depth)
    return compute_edge_metrics([edges[p] for p in range(edges.shape[0])], gt_edge, edge_to_edge_thresh)
