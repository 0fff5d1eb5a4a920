"""GPU (-m gpu): chamfer edge metrics on device (edge half) through the C ABI
(mte_chamfer_distance) against the fixtures produced by the reference's chamfer_distance and against the oracle /
scipy at KITTI size."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import edge_oracle as eo

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[8:-4] for p in glob.glob(os.path.join(GOLDEN, "chamfer_*.npz")))


def close(a, b, rel=1e-12):
    a = float(a)
    if np.isnan(b):
        assert np.isnan(a)
    else:
        assert abs(a - b) <= rel * max(1.0, abs(b)), (a, b)


@pytest.mark.parametrize("name", CASES)
def test_fixtures(name):
    from mindtheedge_amd.utils.edge import chamfer_distance, edge_precision_recall_f1
    z = np.load(os.path.join(GOLDEN, "chamfer_%s.npz" % name))
    pred, gt = torch.from_numpy(z["pred"]).cuda(), torch.from_numpy(z["gt"]).cuda()       # uint8 images
    c, p, m = chamfer_distance(pred, gt)
    assert c.is_cuda and c.dtype == torch.float64
    close(c, float(z["c_pg"])); close(p, float(z["p_pg"]))
    np.testing.assert_array_equal(m.cpu().numpy(), z["m_pg"])
    if "c_gp" in z:
        c, p, m = chamfer_distance(gt, pred)
        close(c, float(z["c_gp"])); close(p, float(z["p_gp"]))
        np.testing.assert_array_equal(m.cpu().numpy(), z["m_gp"])
        c, p, _ = chamfer_distance(pred, gt, edge_to_edge_thresh=2.5, return_map=False)
        close(c, float(z["c_t25"])); close(p, float(z["p_t25"]))
        c, p, _ = chamfer_distance(pred, torch.from_numpy(z["soft_gt"]).cuda())
        close(c, float(z["c_soft"])); close(p, float(z["p_soft"]))
        pr, rc, f1 = edge_precision_recall_f1(pred, gt)
        close(pr, float(z["p_pg"])); close(rc, float(z["p_gp"]))


def _strokes(H, W, n, seed, shift=0.0):
    g = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    im = np.zeros((H, W), bool)
    for _ in range(n):
        cx, cy, rad = g.random() * W + shift, g.random() * H + shift / 3, 5 + g.random() * 120
        im |= np.abs(np.sqrt((x - cx) ** 2 + (y - cy) ** 2) - rad) < 0.6
    return (im * 255).astype(np.uint8)


def test_kitti_size_batch_matches_oracle_and_scipy():
    from scipy import ndimage
    from mindtheedge_amd.utils.edge import chamfer_distance, distance_transform_edt
    B, H, W = 3, 375, 1242
    gt = np.stack([_strokes(H, W, 12, 10 + b) for b in range(B)])
    pred = np.stack([_strokes(H, W, 12, 10 + b, shift=3.0 + 4 * b) for b in range(B)])
    c, p, m = chamfer_distance(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda())
    assert c.shape == (B,) and m.shape == (B, H, W)
    d = distance_transform_edt(torch.from_numpy(gt).cuda()).cpu().numpy()
    for b in range(B):
        want = ndimage.distance_transform_edt(1 - (gt[b] > 127).astype(np.uint8))             # the reference's own call
        np.testing.assert_array_equal(d[b], want.astype(np.float32))
        wc, wp, wm = eo.chamfer_distance(pred[b], gt[b])
        close(c[b], wc); close(p[b], wp)
        np.testing.assert_array_equal(m[b].cpu().numpy(), wm)
    assert 0.0 < float(p[2]) < float(p[0]) <= 1.0                                              # larger displacement, fewer matches


def test_properties():
    from mindtheedge_amd.utils.edge import chamfer_distance, edge_precision_recall_f1
    im = torch.from_numpy(_strokes(96, 320, 6, 3)).cuda()
    c, p, _ = chamfer_distance(im, im)
    assert float(c) == 0.0 and float(p) == 1.0                                                 # identical images
    pr, rc, f1 = edge_precision_recall_f1(im, im)
    assert float(f1) == 1.0
    one = torch.zeros(64, 200, device="cuda")
    one[10, 20] = 255.0
    far = torch.zeros(64, 200, device="cuda")
    far[50, 170] = 255.0
    c, p, _ = chamfer_distance(far, one)
    close(c, float(np.sqrt(40.0 ** 2 + 150.0 ** 2))); assert float(p) == 0.0                   # a single pair: exact distance
    c, p, _ = chamfer_distance(torch.zeros(8, 8, device="cuda"), one[:8, :8] + 255.0)
    assert np.isnan(float(c)) and np.isnan(float(p))                                           # nothing predicted


def test_errors():
    from mindtheedge_amd.utils.edge import chamfer_distance
    from mindtheedge_amd.kernels import MteError
    with pytest.raises(MteError):
        chamfer_distance(torch.zeros(4, 4), torch.zeros(4, 4))
    with pytest.raises(NotImplementedError):
        chamfer_distance(torch.zeros(4, 4).cuda(), torch.zeros(4, 4).cuda(), mask=torch.ones(4, 4))
    with pytest.raises(ValueError):
        chamfer_distance(torch.zeros(4, 4).cuda(), torch.zeros(4, 5).cuda())
