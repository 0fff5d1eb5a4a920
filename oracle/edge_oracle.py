"""CPU restatement of the chamfer edge metrics -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(mindtheedge_amd/) never does.  SURVEY.md 8 row f-3, edge half without the Canny step.

Follows (reference file:line):
  chamfer_distance          packnet_code/packnet_sfm/utils/edge.py:19-64 (same body as /root/reference/edge.py:29-71)
  precision / recall / F1   packnet_code/packnet_sfm/models/model_wrapper.py:426-440

Pinned by tests/golden/chamfer_*.npz, produced by the reference function itself (tests/golden/make_golden_chamfer.py;
scipy.ndimage is the reference's real dependency and is present in this image).  The Euclidean distance transform is
restated from its definition (exact integer squared distances, sqrt in double), not by calling scipy, so the oracle is an
independent check of the kernels' algorithm.  The Canny step that produces the predicted edge image from a depth map
(cv2.Canny, model_wrapper.py:396-400) is NOT restated: OpenCV is absent and its arithmetic cannot be pinned here.
"""
import numpy as np

INF = 1 << 30


def binarise(im):
    """edge.py:30-32 / :38-40: im/255 > 0.5 in float64."""
    return (np.asarray(im, np.float64) / 255 > 0.5)


def squared_edt(zero_mask):
    """Exact squared Euclidean distance to the nearest True pixel of ``zero_mask`` (int64; INF where there is none)."""
    H, W = zero_mask.shape
    g = np.full((H, W), INF, np.int64)
    ys = np.arange(H)
    for x in range(W):
        rows = ys[zero_mask[:, x]]
        if len(rows):
            g[:, x] = np.min(np.abs(ys[:, None] - rows[None, :]), axis=1)
    g2 = np.where(g >= INF, INF, g * g)
    xs = np.arange(W)
    dx2 = (xs[:, None] - xs[None, :]) ** 2                    # [x][x']
    out = np.empty((H, W), np.int64)
    for y in range(H):
        out[y] = np.min(dx2 + g2[y][None, :], axis=1)
    return out


def chamfer_distance(im_pred, im_gt, edge_to_edge_thresh=5):
    """-> (mean distance from predicted edge pixels to the nearest ground-truth edge pixel, fraction of predicted edge
    pixels closer than the threshold, map: -1 off the predicted edges, else 0/1).  mask=None only."""
    gt = binarise(im_gt)
    pred = binarise(im_pred)
    d2 = squared_edt(gt)
    dist = np.sqrt(d2.astype(np.float64))
    n = pred.sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        c_dist = np.float64(dist[pred].sum()) / np.float64(n)
        close = dist < edge_to_edge_thresh
        percentage = np.float64((close & pred).sum()) / np.float64(n)
    cond = np.where(pred, close.astype(np.float64), -1.0)
    return c_dist, percentage, cond


def precision_recall_f1(im_pred, im_gt, edge_to_edge_thresh=5):
    """model_wrapper.py:431-438."""
    _, p, _ = chamfer_distance(im_pred, im_gt, edge_to_edge_thresh)
    _, r, _ = chamfer_distance(im_gt, im_pred, edge_to_edge_thresh)
    with np.errstate(divide="ignore", invalid="ignore"):
        return p, r, 2 * ((p * r) / (p + r))
