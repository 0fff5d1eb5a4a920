"""CPU restatement of the training-target preparation -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(mindtheedge_amd/) never does.  SURVEY.md 8 row f-4 (data half).  Pinned by tests/golden/data_*.npz, produced by the
reference's own ``resize_depth_preserve`` (packnet_sfm/datasets/augmentations.py:58-100) and by the literal expressions of
gta_dataset.py:407-409 / augmentations.py:186-188 (tests/golden/make_golden_data.py)."""
import numpy as np


def resize_depth_preserve(depth, shape):
    depth = np.squeeze(np.asarray(depth))
    h, w = depth.shape
    H, W = shape
    out = np.zeros((H, W))
    ys, xs = np.nonzero(depth > 0)                       # raster order
    ty = (ys * (H / h)).astype(np.int32)
    tx = (xs * (W / w)).astype(np.int32)
    ok = (ty < H) & (tx < W)
    for y, x, v in zip(ty[ok], tx[ok], depth[ys[ok], xs[ok]]):      # the last assignment wins
        out[y, x] = v
    return out


def normal_from_u8(v):
    return (360. * (np.asarray(v) / 255.) - 180) * (np.pi / 180)


def edge_from_u8(v):
    v = np.asarray(v).astype(np.float64)
    return v / 255 if v.max() > 1 else v
