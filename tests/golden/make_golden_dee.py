"""Generates tests/golden/dee_*.npz from the upstream reference (development container only).

Row f-2 of SURVEY.md 8: packnet_code/packnet_sfm/utils/tools.py ``hysteresis`` / ``DFS`` run exactly as they are
(pure numpy loops) -> pinned.  ``non_max_suppression`` calls cv2.Sobel, and OpenCV is not in this image: the
reference's loop is run with ``cv2.Sobel`` bound to oracle.dee_oracle.sobel5 (a restatement of OpenCV's published 5x5
Sobel), so these fixtures pin the NMS decision logic GIVEN the gradients; the Sobel arithmetic itself stays unpinned
(stated in oracle/dee_oracle.py and DESIGN.md).  The Sobel responses used are stored next to the outputs.

    python tests/golden/make_golden_dee.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_import                                   # noqa: E402
from oracle import packnet_oracle as po             # noqa: E402
from oracle import dee_oracle as do                 # noqa: E402


def unit(name, shape):
    return (po.fixture_tensor("dee:" + name, shape) * 0.5 + 0.5).numpy()       # U[0,1) float32


def edge_like(name, H, W):
    """A probability map with a few smooth ridges (so NMS / hysteresis have chains to follow) plus noise."""
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    r = unit(name + ":p", (6,))
    m = np.zeros((H, W), np.float32)
    for k in range(3):
        cx, cy, rad = r[2 * k] * W, r[2 * k + 1] * H, 4.0 + 5.0 * k
        d = np.abs(np.sqrt((x - cx) ** 2 + (y - cy) ** 2) - rad)
        m = np.maximum(m, np.exp(-0.5 * (d / 1.2) ** 2).astype(np.float32))
    ramp = np.exp(-0.5 * ((x - y * 1.7 - 3) / 1.5) ** 2).astype(np.float32) * (0.35 + 0.6 * x / W)
    m = np.maximum(m * (0.4 + 0.6 * unit(name + ":a", (H, W))), ramp)
    return np.clip(m + 0.25 * unit(name + ":n", (H, W)) ** 3, 0, 1).astype(np.float32)


def main():
    assert ref_import.reference_available(), "run in the development container (needs /root/reference)"
    ref_import.install_stubs()
    import cv2
    cv2.CV_64F = 6
    cv2.Sobel = lambda img, ddepth, dx, dy, ksize=5: do.sobel5(img, dx, dy)         # see the module docstring
    from packnet_code.packnet_sfm.utils import tools
    for name, H, W in [("a", 24, 40), ("b", 17, 23), ("c", 32, 64), ("tiny", 3, 3), ("thin", 5, 2)]:
        p = edge_like(name, H, W) if min(H, W) > 3 else unit(name + ":raw", (H, W))
        out = {"prob": p, "sobelx": do.sobel5(p, 1, 0), "sobely": do.sobel5(p, 0, 1)}
        nms = tools.non_max_suppression(p.copy())
        out["nms"] = nms
        out["nms_hyst"] = tools.hysteresis(nms.copy())                               # the shipped chain (float64 input)
        out["hyst_only"] = tools.hysteresis(p.astype(np.float64))                    # frame keeps its values: quirk path
        out["hyst_custom"] = tools.hysteresis(nms.copy(), t_low=0.1, t_high=0.5)
        np.savez_compressed(os.path.join(HERE, "dee_%s.npz" % name), **out)
        print(name, p.shape, "nms kept", int((nms > 0).sum()), "after hysteresis", int((np.nan_to_num(out["nms_hyst"]) > 0).sum()),
              "nan" if np.isnan(out["nms_hyst"]).any() else "")
    # no strong pixel at all: 0/0 -> NaN map (tools.py:82)
    p = (0.5 * edge_like("weak", 12, 20)).astype(np.float32)
    nms = tools.non_max_suppression(p.copy())
    np.savez_compressed(os.path.join(HERE, "dee_nostrong.npz"), prob=p, nms=nms, nms_hyst=tools.hysteresis(nms.copy()),
                        hyst_only=tools.hysteresis(p.astype(np.float64)), sobelx=do.sobel5(p, 1, 0), sobely=do.sobel5(p, 0, 1),
                        hyst_custom=tools.hysteresis(nms.copy(), t_low=0.1, t_high=0.45))
    # a long snake: one strong end, weak body crossing the whole map (needs many raster sweeps in the reference)
    H, W = 20, 48
    s = np.zeros((H, W), np.float64)
    for row in range(2, H - 2, 4):
        s[row, 2:W - 2] = 0.5
        s[row + 1:row + 4, (W - 3) if (row // 4) % 2 == 0 else 2] = 0.5
    s[2, 2] = 0.9
    s[H - 3, 5:9] = 0.6                                                              # disconnected weak piece: must vanish
    np.savez_compressed(os.path.join(HERE, "dee_snake.npz"), img=s, hyst=tools.hysteresis(s.copy()))
    print("snake kept", int((tools.hysteresis(s.copy()) > 0).sum()), "of", int((s > 0).sum()))


if __name__ == "__main__":
    main()
