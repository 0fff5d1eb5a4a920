"""CPU restatement of the Canny step of the validation edge metrics -- TEST INFRASTRUCTURE ONLY.  **Parity unpinned.**

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(mindtheedge_amd/) never does.  SURVEY.md 8 row f-3: ModelWrapper.compute_edge_metrics
(packnet_code/packnet_sfm/models/model_wrapper.py:376-400) turns the predicted depth into three edge images with

    depth_resized_vis = (depth * (255.0 / depth.max())).astype(np.uint8)
    cv2.Canny(depth_resized_vis, 10, 20), cv2.Canny(.., 20, 40), cv2.Canny(.., 30, 60)

OpenCV is a third-party dependency that is absent from this image (the reference pins no version), so nothing here can
be checked against the real cv2.Canny: the functions below restate OpenCV's published algorithm (imgproc/src/canny.cpp,
3.x/4.x, apertureSize = 3, L2gradient = false) and are held to known-answer tests only:
  * dx, dy = 3x3 Sobel on the uint8 image, BORDER_REPLICATE, 16-bit signed;  magnitude = |dx| + |dy|
  * a pixel is a candidate iff magnitude > low and it is a local maximum along its gradient sector, sectors split at
    tan(22.5 deg) and tan(67.5 deg) in 15-bit fixed point (TG22 = round(0.41421356 * 2^15)):
      horizontal  m >  left  and m >= right        vertical  m >  up  and m >= down
      diagonal    m >  both neighbours on the diagonal chosen by the sign of dx*dy          (magnitude outside the image = 0)
  * candidates with magnitude > high are edges; other candidates become edges when 8-connected to an edge
  * output 255 / 0
"""
import numpy as np

TG22 = int(0.4142135623730950488016887242097 * (1 << 15) + 0.5)


def resize_linear(img, W, H):
    """cv2.resize(img, (W, H), interpolation=cv2.INTER_LINEAR) on a float32 image (model_wrapper.py:386-387), restated from
    OpenCV's resize.cpp: half-pixel centres, fractions zeroed where the source index is clamped, rows first then columns."""
    src = np.asarray(img, np.float32)
    h, w = src.shape

    def taps(n_dst, n_src):
        scale = np.float32(np.float64(n_src) / n_dst)
        f = ((np.arange(n_dst) + 0.5) * np.float64(scale) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        s[lo], f[lo] = 0, 0
        hi = s >= n_src - 1
        s[hi], f[hi] = n_src - 1, 0
        return s, np.minimum(s + 1, n_src - 1), (np.float32(1) - f).astype(np.float32), f
    x0, x1, a0, a1 = taps(W, w)
    y0, y1, b0, b1 = taps(H, h)
    rows = (src[:, x0] * a0[None, :] + src[:, x1] * a1[None, :]).astype(np.float32)
    return (rows[y0] * b0[:, None] + rows[y1] * b1[:, None]).astype(np.float32)


def depth_to_u8(depth):
    """model_wrapper.py:396-397 (float32 array times a float32 factor, truncated)."""
    d = np.asarray(depth, np.float32)
    factor = np.float32(255.0) / np.float32(d.max())
    return (d * factor).astype(np.uint8)


def sobel3_s16(img):
    p = np.pad(np.asarray(img, np.int32), 1, mode="edge")                 # BORDER_REPLICATE
    H, W = img.shape
    s = lambda dy, dx: p[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]            # noqa: E731
    dx = (s(-1, 1) + 2 * s(0, 1) + s(1, 1)) - (s(-1, -1) + 2 * s(0, -1) + s(1, -1))
    dy = (s(1, -1) + 2 * s(1, 0) + s(1, 1)) - (s(-1, -1) + 2 * s(-1, 0) + s(-1, 1))
    return dx, dy


def local_maxima(dx, dy):
    """-> (magnitude, boolean map of pixels that survive non-maximum suppression; thresholds not applied)."""
    H, W = dx.shape
    mag = np.abs(dx) + np.abs(dy)
    mp = np.pad(mag, 1)                                                   # zero outside the image
    n = lambda oy, ox: mp[1 + oy:1 + oy + H, 1 + ox:1 + ox + W]           # noqa: E731
    x, y = np.abs(dx).astype(np.int64), np.abs(dy).astype(np.int64) << 15
    tg22x = x * TG22
    tg67x = tg22x + (x << 16)
    horiz = y < tg22x
    vert = ~horiz & (y > tg67x)
    diag = ~horiz & ~vert
    s_pos = (dx ^ dy) >= 0                                                # same sign: the '\\' diagonal
    keep_h = (mag > n(0, -1)) & (mag >= n(0, 1))
    keep_v = (mag > n(-1, 0)) & (mag >= n(1, 0))
    keep_d = np.where(s_pos, (mag > n(-1, -1)) & (mag > n(1, 1)), (mag > n(-1, 1)) & (mag > n(1, -1)))
    return mag, (horiz & keep_h) | (vert & keep_v) | (diag & keep_d)


def canny(img_u8, threshold1, threshold2):
    from scipy import ndimage
    low, high = int(min(threshold1, threshold2)), int(max(threshold1, threshold2))
    dx, dy = sobel3_s16(img_u8)
    mag, is_max = local_maxima(dx, dy)
    cand = is_max & (mag > low)
    strong = cand & (mag > high)
    lab, n = ndimage.label(cand, structure=np.ones((3, 3), int))
    keep = np.zeros(n + 1, bool)
    keep[np.unique(lab[strong])] = True
    keep[0] = False
    return (keep[lab] * 255).astype(np.uint8)


def edges_from_depth(depth, thresholds=((10, 20), (20, 40), (30, 60))):
    u8 = depth_to_u8(depth)
    return [canny(u8, a, b) for a, b in thresholds]
