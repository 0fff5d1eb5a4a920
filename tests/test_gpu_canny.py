"""GPU (-m gpu): the Canny step of the validation edge metrics (mte_canny_*) against oracle/canny_oracle.py.
PARITY UNPINNED -- both sides restate OpenCV's published algorithm; cv2 itself is not available (see the oracle header).
What these tests establish is that the kernels and the independent numpy restatement agree bit for bit, plus known answers."""
import numpy as np
import pytest
import torch

from oracle import canny_oracle as co
from oracle import edge_oracle as eo

pytestmark = pytest.mark.gpu


def _depth(B, H, W, seed):
    """piecewise-planar scene with a few objects and a little noise (metres)"""
    g = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    out = []
    for _ in range(B):
        d = 5.0 + 60.0 * (1.0 - y / H) ** 2 + 0.01 * x
        for _ in range(12):
            cx, cy, r, z = g.random() * W, g.random() * H, 6 + g.random() * 50, 4 + g.random() * 40
            m = ((x - cx) / r) ** 2 + ((y - cy) / (0.6 * r)) ** 2 < 1
            d = np.where(m, np.minimum(d, z + 0.02 * (x - cx)), d)
        out.append((d + 0.05 * g.standard_normal((H, W))).astype(np.float32))
    return np.stack(out)


@pytest.mark.parametrize("B,H,W", [(1, 24, 40), (2, 96, 320), (2, 375, 1242), (1, 5, 3), (1, 1, 9)])
def test_edges_match_restatement(B, H, W):
    from mindtheedge_amd.utils.edge import canny_from_depth
    d = _depth(B, H, W, seed=H + W)
    edges, vis = canny_from_depth(torch.from_numpy(d).cuda(), return_vis=True)
    assert edges.shape == (3, B, H, W) and edges.dtype == torch.float32
    for b in range(B):
        np.testing.assert_array_equal(vis[b].cpu().numpy(), co.depth_to_u8(d[b]))
        want = co.edges_from_depth(d[b])
        for p in range(3):
            np.testing.assert_array_equal(edges[p, b].cpu().numpy(), want[p].astype(np.float32))
    if H > 50:
        n = [int((edges[p] > 0).sum()) for p in range(3)]
        assert n[0] >= n[1] >= n[2] > 0                      # higher thresholds keep fewer edges


def test_known_answers():
    from mindtheedge_amd.utils.edge import canny_from_depth
    step = np.ones((12, 12), np.float32)
    step[:, 6:] = 2.55                                       # uint8: 100 | 255
    e = canny_from_depth(torch.from_numpy(step).cuda(), thresholds=((10, 20),))[0].cpu().numpy()
    assert np.all(e[:, 5] == 255) and e.sum() == 255 * 12   # one-pixel line on the dark side of the step ('>' left, '>=' right)
    flat = np.full((8, 8), 3.0, np.float32)
    assert float(canny_from_depth(torch.from_numpy(flat).cuda()).abs().sum()) == 0.0
    # a weak ridge attached to a strong one survives, a detached weak ridge does not
    img = np.full((20, 40), 1.0, np.float32)
    img[:, 10:] = 1.12                                        # weak step (uint8 difference 12*... see below)
    img[0:10, 10:] = 2.0                                      # upper half: strong step at the same column
    img[:, 30:] = np.where(np.arange(20)[:, None] >= 12, img[:, 30:] + 0.05, img[:, 30:])
    e = canny_from_depth(torch.from_numpy(img).cuda(), thresholds=((20, 200),))[0].cpu().numpy()
    want = co.canny(co.depth_to_u8(img), 20, 200)
    np.testing.assert_array_equal(e, want.astype(np.float32))


def test_edge_metrics_from_depth_chain():
    from mindtheedge_amd.utils.edge import compute_edge_metrics_from_depth
    d = _depth(1, 192, 640, seed=3)[0]
    gt = co.canny(co.depth_to_u8(_depth(1, 192, 640, seed=3)[0] * 1.0), 20, 40)          # ground truth = the middle setting itself
    got = compute_edge_metrics_from_depth(torch.from_numpy(d).cuda(), torch.from_numpy(gt).cuda()).cpu().numpy()
    want = []
    for e in co.edges_from_depth(d):
        want.extend(eo.precision_recall_f1(e, gt))
    np.testing.assert_allclose(got, np.array(want, np.float64), rtol=1e-12)
    assert got[3] == 1.0 and got[4] == 1.0 and got[5] == 1.0                             # identical edge images for (20, 40)
    # KITTI-like: 192x640 prediction against a 187x621 edge image -> resized first (cv2.resize INTER_LINEAR restatement)
    gt2 = co.canny(co.depth_to_u8(co.resize_linear(d, 621, 187)), 20, 40)
    got2 = compute_edge_metrics_from_depth(torch.from_numpy(d).cuda(), torch.from_numpy(gt2).cuda()).cpu().numpy()
    want2 = []
    for e in co.edges_from_depth(co.resize_linear(d, 621, 187)):
        want2.extend(eo.precision_recall_f1(e, gt2))
    np.testing.assert_allclose(got2, np.array(want2, np.float64), rtol=1e-12)


@pytest.mark.parametrize("h,w,H,W", [(384, 1280, 375, 1242), (12, 20, 31, 47), (9, 9, 9, 9), (5, 7, 1, 1), (1, 1, 4, 6), (40, 64, 20, 32)])
def test_resize_linear_matches_restatement(h, w, H, W):
    from mindtheedge_amd.utils.edge import resize_linear
    g = np.random.default_rng(h * 7 + W)
    src = (g.random((h, w)) * 80).astype(np.float32)
    got = resize_linear(torch.from_numpy(src).cuda(), (H, W)).cpu().numpy()
    np.testing.assert_array_equal(got, co.resize_linear(src, W, H))
    if (h, w) == (H, W):
        np.testing.assert_array_equal(got, src)
    if (h, w, H, W) == (40, 64, 20, 32):                     # exact 2x reduction: every output is the mean of a 2x2 block
        np.testing.assert_allclose(got, src.reshape(20, 2, 32, 2).mean((1, 3)), rtol=1e-6)
