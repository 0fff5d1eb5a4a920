"""Generates tests/golden/chamfer_*.npz from the upstream reference (development container only):
packnet_code/packnet_sfm/utils/edge.py ``chamfer_distance`` run as it is (its scipy.ndimage dependency is installed).

    python tests/golden/make_golden_chamfer.py
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
    return (po.fixture_tensor("chamfer:" + name, shape) * 0.5 + 0.5).numpy()


def curves(name, H, W, n, jitter):
    """uint8 edge image (0/255) of n random circles/lines; ``jitter`` displaces a second copy for the prediction."""
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    r = unit(name, (n, 3))
    im = np.zeros((H, W), bool)
    for k in range(n):
        cx, cy, rad = r[k, 0] * W + jitter, r[k, 1] * H - jitter / 2, 3 + r[k, 2] * min(H, W) / 3
        im |= np.abs(np.sqrt((x - cx) ** 2 + (y - cy) ** 2) - rad) < 0.6
    return (im * 255).astype(np.uint8)


def main():
    assert ref_import.reference_available(), "run in the development container (needs /root/reference)"
    ref_import.install_stubs()
    from packnet_code.packnet_sfm.utils.edge import chamfer_distance
    for name, H, W, n, jitter in [("a", 40, 64, 4, 2.0), ("b", 33, 47, 3, 6.0), ("far", 40, 160, 1, 30.0), ("same", 30, 30, 3, 0.0)]:
        gt = curves(name, H, W, n, 0.0)
        pred = curves(name, H, W, n, jitter)
        if name == "far":                                    # two distant strokes: nothing within the threshold
            gt, pred = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
            gt[5:35, 20] = 255
            pred[8:30, 20 + int(jitter)] = 255
            pred[3, 100:150] = 255
        out = {"pred": pred, "gt": gt}
        for tag, a, b in (("pg", pred, gt), ("gp", gt, pred)):
            c, p, m = chamfer_distance(a.astype(np.float64), b.astype(np.float64))
            out["c_" + tag], out["p_" + tag], out["m_" + tag] = np.float64(c), np.float64(p), m
        c, p, _ = chamfer_distance(pred.astype(np.float64), gt.astype(np.float64), edge_to_edge_thresh=2.5)
        out["c_t25"], out["p_t25"] = np.float64(c), np.float64(p)
        # soft inputs: anything above 127.5 counts (edge.py:30-32)
        soft = (gt.astype(np.float64) * (0.3 + 0.7 * unit(name + ":soft", (H, W))))
        c, p, _ = chamfer_distance(pred.astype(np.float64), soft.copy())
        out["soft_gt"], out["c_soft"], out["p_soft"] = soft, np.float64(c), np.float64(p)
        np.savez_compressed(os.path.join(HERE, "chamfer_%s.npz" % name), **out)
        print(name, "c_dist %.4f perc %.4f | reverse %.4f %.4f" % (out["c_pg"], out["p_pg"], out["c_gp"], out["p_gp"]))
    # no predicted edge pixel at all: 0/0
    gt = curves("a", 40, 64, 4, 0.0)
    c, p, m = chamfer_distance(np.zeros((40, 64)), gt.astype(np.float64))
    np.savez_compressed(os.path.join(HERE, "chamfer_nopred.npz"), pred=np.zeros((40, 64), np.uint8), gt=gt, c_pg=np.float64(c), p_pg=np.float64(p), m_pg=m)
    print("nopred", c, p)


if __name__ == "__main__":
    main()
