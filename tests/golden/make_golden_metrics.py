"""Generates tests/golden/metrics_*.npz from the upstream reference (development container only).

Row f-3 of SURVEY.md 8: flip-TTA ``post_process_inv_depth`` and ``compute_depth_metrics``
(packnet_code/packnet_sfm/utils/depth.py:230-325).  Inputs come from the same deterministic generator as every other
fixture (oracle.packnet_oracle.fixture_tensor); outputs are what the reference functions return on them.

    python tests/golden/make_golden_metrics.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_import                                   # noqa: E402
from oracle import packnet_oracle as po             # noqa: E402


def unit(name, shape):
    return po.fixture_tensor("metrics:" + name, shape) * 0.5 + 0.5          # U[0,1)


def depth_pair(name, B, H, W, h, w, holes=0.35):
    """LiDAR-like ground truth (zeros where there is no return) and a dense prediction around it."""
    gt = 1.0 + 90.0 * unit(name + ":gt", (B, 1, H, W)) ** 2                 # up to ~91 m: some beyond max_depth
    gt = gt * (unit(name + ":holes", (B, 1, H, W)) > holes).float()
    pred = 0.5 + 60.0 * unit(name + ":pred", (B, 1, h, w))
    return gt, pred


CASES = [   # name, B, H, W, h, w, crop, scale_output, min_depth, max_depth
    ("resize_garg", 2, 37, 124, 40, 128, "garg", "resize", 0.0, 80.0),
    ("resize_nocrop", 1, 30, 50, 16, 24, "", "resize", 1.5, 60.0),
    ("same_garg", 2, 24, 80, 24, 80, "garg", "resize", 0.0, 80.0),
    ("topcenter", 2, 37, 124, 32, 120, "garg", "top-center", 0.0, 80.0),
    ("empty_image", 3, 20, 48, 20, 48, "", "resize", 0.0, 80.0),
    ("even_count", 1, 4, 6, 4, 6, "", "resize", 0.0, 1000.0),
]


def main():
    assert ref_import.reference_available(), "run in the development container (needs /root/reference)"
    ref_import.install_stubs()
    from packnet_code.packnet_sfm.utils.depth import compute_depth_metrics, post_process_inv_depth
    torch.set_num_threads(1)
    for name, B, H, W, h, w, crop, scale_output, dmin, dmax in CASES:
        gt, pred = depth_pair(name, B, H, W, h, w)
        if name == "empty_image":
            gt[1] = 0.0                                                     # the reference skips it but still divides by B
        if name == "even_count":
            gt = 1.0 + 90.0 * unit(name + ":gt", (B, 1, H, W))              # all 24 pixels valid: lower median of an even count
        cfg = types.SimpleNamespace(crop=crop, scale_output=scale_output, min_depth=dmin, max_depth=dmax)
        out = {"gt": gt, "pred": pred, "crop": np.array(crop), "scale_output": np.array(scale_output),
               "min_depth": np.float64(dmin), "max_depth": np.float64(dmax)}
        for use_gt_scale in (False, True):
            m = compute_depth_metrics(cfg, gt.clone(), pred.clone(), use_gt_scale=use_gt_scale)
            out["metrics_gt%d" % int(use_gt_scale)] = m
        np.savez_compressed(os.path.join(HERE, "metrics_%s.npz" % name),
                            **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
        print(name, out["metrics_gt0"].numpy(), out["metrics_gt1"].numpy())
    # flip-TTA fusion
    B, H, W = 2, 12, 40
    inv = 0.01 + unit("pp:inv", (B, 1, H, W))
    inv_f = 0.01 + unit("pp:invf", (B, 1, H, W))
    out = {"inv_depth": inv, "inv_depth_flipped": inv_f}
    for method in ("mean", "max", "min"):
        out["pp_" + method] = post_process_inv_depth(inv.clone(), inv_f.clone(), method=method)
    np.savez_compressed(os.path.join(HERE, "metrics_post_process.npz"), **{k: v.numpy() for k, v in out.items()})
    print("post_process", {k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
