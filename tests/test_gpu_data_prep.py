"""GPU (-m gpu): training-target preparation on device (data half) through the C ABI against the
fixtures produced by the reference (tests/golden/make_golden_data.py) and the oracle at KITTI size."""
import os

import numpy as np
import pytest
import torch

from oracle import data_oracle as do

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data_prep.npz")


def test_fixtures():
    from mindtheedge_amd.datasets.kitti_edges import edge_target, normal_target, resize_depth_preserve
    z = np.load(GOLDEN)
    for name in ("down", "kitti", "up", "same", "empty"):
        got = resize_depth_preserve(torch.from_numpy(z["rdp_%s_in" % name]).cuda(), tuple(int(v) for v in z["rdp_%s_shape" % name]))
        np.testing.assert_array_equal(got.cpu().numpy(), z["rdp_%s_out" % name].astype(np.float32))
    got = resize_depth_preserve(torch.from_numpy(z["rdp_down_in"]).cuda(), 0.5)
    np.testing.assert_array_equal(got.cpu().numpy(), z["rdp_ratio_out"].astype(np.float32))
    u8 = torch.from_numpy(z["u8"]).cuda()
    np.testing.assert_array_equal(normal_target(u8).cpu().numpy(), z["normal"].astype(np.float32))
    np.testing.assert_array_equal(edge_target(u8).cpu().numpy(), z["edge"].astype(np.float32))
    assert torch.equal(edge_target((u8 > 128).to(torch.uint8)), (u8 > 128).float())          # a 0/1 map is left alone (max <= 1)


def test_kitti_size_batch_against_oracle():
    from mindtheedge_amd.datasets.kitti_edges import resize_depth_preserve
    g = np.random.default_rng(0)
    d = ((g.random((3, 375, 1242)) < 0.05) * (1 + 80 * g.random((3, 375, 1242)))).astype(np.float32)
    got = resize_depth_preserve(torch.from_numpy(d).cuda(), (384, 1280)).cpu().numpy()
    for b in range(3):
        np.testing.assert_array_equal(got[b], do.resize_depth_preserve(d[b], (384, 1280)).astype(np.float32))
    down = resize_depth_preserve(torch.from_numpy(d).cuda(), (96, 320)).cpu().numpy()         # heavy collisions: last in raster order wins
    for b in range(3):
        np.testing.assert_array_equal(down[b], do.resize_depth_preserve(d[b], (96, 320)).astype(np.float32))


def test_prepare_edge_sample_shapes_and_errors():
    from mindtheedge_amd.datasets.kitti_edges import prepare_edge_sample, normal_target
    from mindtheedge_amd.kernels import MteError
    g = torch.Generator().manual_seed(1)
    edges = [((torch.rand(384 >> s, 1280 >> s, generator=g) < 0.03) * 255).to(torch.uint8).cuda() for s in range(4)]
    normals = [torch.randint(0, 256, (384 >> s, 1280 >> s), generator=g, dtype=torch.uint8).cuda() for s in range(4)]
    out = prepare_edge_sample(edges, normals, (384, 1280))
    assert set(out) == {"edge", "edge_1", "edge_2", "edge_3", "normal", "normal_1", "normal_2", "normal_3"}
    for s in range(4):
        sfx = "" if s == 0 else "_%d" % s
        assert out["edge" + sfx].shape == (1, 384 >> s, 1280 >> s) and float(out["edge" + sfx].max()) == 1.0
        assert torch.equal(out["edge" + sfx][0], edges[s].float() / 255.0)
        assert out["normal" + sfx].shape == (1, 384 >> s, 1280 >> s)
        assert float(out["normal" + sfx].min()) >= -np.pi - 1e-6 and float(out["normal" + sfx].max()) <= np.pi + 1e-6
    with pytest.raises(MteError):
        normal_target(torch.zeros(4, 4, dtype=torch.uint8))
    with pytest.raises(ValueError):
        normal_target(torch.zeros(4, 4).cuda())
