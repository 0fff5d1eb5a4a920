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


def test_split_dataset_end_to_end(tmp_path):
    """PNG files + an 8-column split file on disk -> device batches with the reference's keys, scales and value ranges,
    then one training step of the model on such a batch."""
    from PIL import Image
    from mindtheedge_amd.datasets.kitti_edges import KittiEdgeSplitDataset, SplitLoader
    H, W = 64, 128
    g = np.random.default_rng(0)
    lines = []
    for i in range(3):
        Image.fromarray((g.random((H + 10, W + 20, 3)) * 255).astype(np.uint8)).save(os.path.join(tmp_path, "rgb%d.png" % i))
        depth = ((g.random((H, W)) < 0.1) * (1 + 79 * g.random((H, W))) * 256).astype(np.uint16)
        depth[0, 0] = 300 * 256 // 256 + 300                                   # keep max > 255
        Image.fromarray(depth).save(os.path.join(tmp_path, "depth%d.png" % i))
        for s in range(4):
            e = ((g.random((H >> s, W >> s)) < 0.05) * 255).astype(np.uint8)
            Image.fromarray(e).save(os.path.join(tmp_path, "%08d_lidar_00%d.png" % (i, s)))
            os.makedirs(os.path.join(tmp_path, "normals"), exist_ok=True)
            Image.fromarray(g.integers(0, 256, (H >> s, W >> s), dtype=np.uint8)).save(os.path.join(tmp_path, "normals", "%08d_lidar_00%d.png" % (i, s)))
        lines.append("rgb%d.png depth%d.png %08d_lidar_000.png depth%d.png None None None normals/%08d_lidar_000.png\n" % (i, i, i, i, i))
    split = os.path.join(tmp_path, "split.txt")
    open(split, "w").writelines(lines)
    ds = KittiEdgeSplitDataset(split, (H, W), root=str(tmp_path))
    assert len(ds) == 3
    s0 = ds[0]
    assert s0["rgb"].shape == (3, H, W) and 0.0 <= float(s0["rgb"].min()) and float(s0["rgb"].max()) <= 1.0
    d_png = np.array(Image.open(os.path.join(tmp_path, "depth0.png")), dtype=int)
    want_d = do.resize_depth_preserve(np.where(d_png == 0, -1.0, d_png / 256.0).astype(np.float32), (H, W)).astype(np.float32)
    np.testing.assert_array_equal(s0["depth"][0].cpu().numpy(), want_d)
    for s in range(4):
        sfx = "" if s == 0 else "_%d" % s
        e_png = np.array(Image.open(os.path.join(tmp_path, "00000000_lidar_00%d.png" % s)))
        np.testing.assert_array_equal(s0["edge" + sfx][0].cpu().numpy(), (e_png / 255.0).astype(np.float32))
        n_png = np.array(Image.open(os.path.join(tmp_path, "normals", "00000000_lidar_00%d.png" % s)))
        np.testing.assert_array_equal(s0["normal" + sfx][0].cpu().numpy(), do.normal_from_u8(n_png).astype(np.float32))
    loader = SplitLoader(ds, 2, shuffle=False)
    batches = list(loader)
    assert len(batches) == 1 and batches[0]["rgb"].shape == (2, 3, H, W) and batches[0]["edge_2"].shape == (2, 1, H // 4, W // 4)
    # a training step on the loaded batch
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    from mindtheedge_amd.utils.config import load_config
    K.set_compute_dtype("bf16")
    cfg = load_config(None, {"model": {"loss": {"supervised_method": "sparse-silog", "supervised_num_scales": 1, "supervised_loss_weight": 1.0,
                                                  "edges_depth_edge_loss_all_scales": True, "flip_lr_prob": 0.0}}})
    wrap = ModelWrapper(cfg).cuda().train()
    out = wrap.training_step(batches[0])
    assert torch.isfinite(out["loss"]).all()
    out["loss"].sum().backward()
    # the train_edges.py --data hook, two ranks' shares are disjoint
    from mindtheedge_amd.datasets.kitti_edges import make_loader
    cfg2 = load_config(None, {"datasets": {"augmentation": {"image_shape": (H, W)}, "train": {"batch_size": 1, "split": [split], "path": [str(tmp_path)]}}})
    l0, l1 = make_loader(cfg2, 0, 2), make_loader(cfg2, 1, 2)
    i0 = [b["idx"][0] for b in l0]
    i1 = [b["idx"][0] for b in l1]
    assert len(i0) == len(i1) == 1 and set(i0).isdisjoint(i1)
