"""CPU restatement of the reference's validation depth metrics -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(mindtheedge_amd/) never does.  SURVEY.md 8 row f-3 (depth half): flip-TTA fusion and the seven depth metrics.

Pinned by tests/golden/metrics_*.npz, generated from the reference itself by tests/golden/make_golden_metrics.py
(tests/test_metrics_oracle.py compares).  numpy, float32 where the reference computes in float32.

Follows (reference file:line, under packnet_code/packnet_sfm/):
  fuse_inv_depth            utils/depth.py:202-227
  post_process_inv_depth    utils/depth.py:230-256
  compute_depth_metrics     utils/depth.py:259-325
  scale_depth               utils/depth.py:328-361   ('resize' = bilinear, align_corners=True, utils/image.py:122-151)
"""
import numpy as np

F = np.float32


def flip_lr(x):
    return x[..., ::-1]


def fuse_inv_depth(inv_depth, inv_depth_hat, method="mean"):
    if method == "mean":
        return F(0.5) * (inv_depth + inv_depth_hat)
    if method == "max":
        return np.maximum(inv_depth, inv_depth_hat)
    if method == "min":
        return np.minimum(inv_depth, inv_depth_hat)
    raise ValueError("Unknown post-process method {}".format(method))


def post_process_inv_depth(inv_depth, inv_depth_flipped, method="mean"):
    """depth.py:230-256: blend the prediction with the un-flipped prediction of the mirrored image; the 5 % border
    columns on either side come from one of the two only (ramp 20*(x-0.05) clamped to [0,1])."""
    inv_depth = np.asarray(inv_depth, F)
    B, C, H, W = inv_depth.shape
    inv_depth_hat = flip_lr(np.asarray(inv_depth_flipped, F))
    fused = fuse_inv_depth(inv_depth, inv_depth_hat, method)
    # torch.linspace(0, 1, W) in float32: start + i*step in the first half, end - (W-1-i)*step in the second
    step = F(1.0) / F(max(W - 1, 1))
    i = np.arange(W)
    xs = np.where(i < W // 2, step * i.astype(F), F(1.0) - step * (W - 1 - i).astype(F)).astype(F)
    mask = F(1.0) - np.clip(F(20.0) * (xs - F(0.05)), F(0.0), F(1.0))
    mask = np.broadcast_to(mask, inv_depth.shape)
    mask_hat = flip_lr(mask)
    return mask_hat * inv_depth + mask * inv_depth_hat + (F(1.0) - mask - mask_hat) * fused


def resize_bilinear_align_corners(x, H, W):
    """F.interpolate(x, (H,W), mode='bilinear', align_corners=True) on [B,1,h,w] float32."""
    x = np.asarray(x, F)
    h, w = x.shape[-2:]
    if (h, w) == (H, W):
        return x
    sy = F(h - 1) / F(H - 1) if H > 1 else F(0.0)
    sx = F(w - 1) / F(W - 1) if W > 1 else F(0.0)
    ry = (sy * np.arange(H, dtype=F)).astype(F)
    rx = (sx * np.arange(W, dtype=F)).astype(F)
    y0 = np.minimum(ry.astype(np.int64), h - 1)
    x0 = np.minimum(rx.astype(np.int64), w - 1)
    y1 = np.minimum(y0 + 1, h - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    ly = (ry - y0.astype(F)).astype(F)[:, None]
    lx = (rx - x0.astype(F)).astype(F)[None, :]
    hy, hx = F(1.0) - ly, F(1.0) - lx
    a = x[..., y0[:, None], x0[None, :]]
    b = x[..., y0[:, None], x1[None, :]]
    c = x[..., y1[:, None], x0[None, :]]
    d = x[..., y1[:, None], x1[None, :]]
    return (hy * (hx * a + lx * b) + ly * (hx * c + lx * d)).astype(F)


def scale_depth(pred, gt_shape, scale_fn):
    pred = np.asarray(pred, F)
    H, W = gt_shape[-2:]
    if scale_fn == "resize":
        return resize_bilinear_align_corners(pred, H, W)
    if scale_fn == "top-center":
        out = np.zeros(tuple(pred.shape[:2]) + (H, W), F)
        top, left = H - pred.shape[2], (W - pred.shape[3]) // 2
        out[:, :, top:top + pred.shape[2], left:left + pred.shape[3]] = pred
        return out
    raise NotImplementedError("Depth scale function {} not implemented.".format(scale_fn))


def lower_median(v):
    """torch.median: the element of rank (n-1)//2 (no averaging for even counts)."""
    s = np.sort(v, kind="stable")
    return s[(len(s) - 1) // 2]


def compute_depth_metrics(gt, pred, crop="", scale_output="resize", min_depth=0.0, max_depth=80.0, use_gt_scale=True):
    """depth.py:259-325 -> float32[7] = (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3), averaged over the batch
    (images without a valid pixel contribute zero but still count in the divisor)."""
    gt = np.asarray(gt, F)
    B, _, H, W = gt.shape
    pred = scale_depth(pred, gt.shape, scale_output)
    crop_mask = None
    if crop == "garg":
        crop_mask = np.zeros((H, W), bool)
        y1, y2 = int(0.40810811 * H), int(0.99189189 * H)
        x1, x2 = int(0.03594771 * W), int(0.96405229 * W)
        crop_mask[y1:y2, x1:x2] = True
    acc = np.zeros(7, np.float64)
    lo, hi = F(min_depth), F(max_depth)
    for i in range(B):
        g, p = gt[i, 0], pred[i, 0]
        valid = (g > lo) & (g < hi)
        if crop_mask is not None:
            valid &= crop_mask
        if not valid.any():
            continue
        g, p = g[valid], p[valid]
        if use_gt_scale:
            p = (p * lower_median(g)) / lower_median(p)
        p = np.clip(p, lo, hi).astype(F)
        with np.errstate(divide="ignore", invalid="ignore"):
            thresh = np.maximum(g / p, p / g)
            d = g - p
            m = lambda v: np.mean(v.astype(np.float64))      # noqa: E731  (float64 mean; the reference's float32 mean agrees to ~1e-6)
            acc += [m(np.abs(d) / g), m(d * d / g), np.sqrt(m(d * d)), np.sqrt(m((np.log(g) - np.log(p)) ** 2)),
                    m(thresh < F(1.25)), m(thresh < F(1.25 ** 2)), m(thresh < F(1.25 ** 3))]
    return (acc / B).astype(F)
