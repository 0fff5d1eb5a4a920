"""Generates tests/golden/data_*.npz from the upstream reference (development container only): ``resize_depth_preserve``
(packnet_code/packnet_sfm/datasets/augmentations.py:58-100) run as it is, and the literal normal de-quantisation /
edge scaling expressions of gta_dataset.py:407-409 and augmentations.py:186-188.

    python tests/golden/make_golden_data.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_import                                   # noqa: E402
from oracle import packnet_oracle as po             # noqa: E402


def unit(name, shape):
    return (po.fixture_tensor("data:" + name, shape) * 0.5 + 0.5).numpy()


def main():
    assert ref_import.reference_available()
    ref_import.install_stubs()
    from packnet_code.packnet_sfm.datasets.augmentations import resize_depth_preserve
    out = {}
    for name, (h, w), (H, W), density in [("down", (37, 124), (12, 40), 0.3), ("kitti", (75, 248), (64, 256), 0.08),
                                           ("up", (9, 14), (20, 33), 0.5), ("same", (16, 24), (16, 24), 0.4), ("empty", (8, 8), (4, 4), 0.0)]:
        d = (unit(name + ":m", (h, w)) < density) * (1.0 + 80.0 * unit(name + ":v", (h, w)))
        d = d.astype(np.float32)
        out["rdp_%s_in" % name] = d
        out["rdp_%s_shape" % name] = np.array([H, W])
        out["rdp_%s_out" % name] = resize_depth_preserve(d.copy(), (H, W))[:, :, 0]
    out["rdp_ratio_out"] = resize_depth_preserve(out["rdp_down_in"].copy(), 0.5)[:, :, 0]
    v = np.arange(256, dtype=np.uint8).reshape(16, 16)
    out["u8"] = v
    out["normal"] = (360. * (v / 255.) - 180) * (np.pi / 180)                  # gta_dataset.py:409
    e = resize_depth_preserve(v.astype(np.float64), (16, 16))                   # augmentations.py:183-188
    if np.max(e) > 1:
        e = e / 255
    out["edge"] = e[:, :, 0]
    np.savez_compressed(os.path.join(HERE, "data_prep.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
