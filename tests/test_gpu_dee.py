"""GPU (-m gpu): depth-edge annotation post-processing on device through the C ABI
(mte_dee_sobel_nms, mte_hysteresis_*) against the fixtures produced by the reference's loops
(tests/golden/make_golden_dee.py) and against the numpy oracle at annotation size."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import dee_oracle as do

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "dee_*.npz")) if "snake" not in p)


def same(got, want, rtol=0.0):
    got = got.double().cpu().numpy() if torch.is_tensor(got) else got
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    if rtol:
        np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(want), rtol=rtol, atol=0)
    else:
        np.testing.assert_array_equal(np.nan_to_num(got), np.nan_to_num(want))


@pytest.mark.parametrize("name", CASES)
def test_fixtures_bit_exact(name):
    from mindtheedge_amd.utils import tools
    z = np.load(os.path.join(GOLDEN, "dee_%s.npz" % name))
    p = torch.from_numpy(z["prob"]).cuda()
    nms = tools.non_max_suppression(p)
    assert nms.is_cuda and nms.dtype == torch.float32 and nms.shape == p.shape
    same(nms, z["nms"])
    same(tools.hysteresis(nms), z["nms_hyst"])
    custom = (0.1, 0.45) if name == "nostrong" else (0.1, 0.5)
    same(tools.hysteresis(nms, *custom), z["hyst_custom"])
    # without NMS the frame keeps value*value/max in float64 upstream; the device map is float32
    same(tools.hysteresis(p), z["hyst_only"], rtol=1e-7)
    n = tools.sobel_normals(p)
    assert n.dtype == torch.uint8
    np.testing.assert_array_equal(n.cpu().numpy(), do.normals_u8(z["prob"]))


def test_snake_converges_over_many_sweeps():
    from mindtheedge_amd.utils import tools
    z = np.load(os.path.join(GOLDEN, "dee_snake.npz"))
    img = torch.from_numpy(z["img"]).float().cuda()
    got = tools.hysteresis(img)
    same(got, z["hyst"], rtol=1e-7)
    assert int((got > 0).sum()) == int((z["img"] > 0).sum()) - 4


def _edge_map(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    m = torch.zeros(B, H, W)
    for b in range(B):
        for _ in range(40):
            cx, cy, rad = torch.rand(3, generator=g) * torch.tensor([W, H, 60.0])
            d = (torch.sqrt((x - cx) ** 2 + (y - cy) ** 2) - (8 + rad)).abs()
            m[b] = torch.maximum(m[b], torch.exp(-0.5 * (d / 1.3) ** 2) * (0.3 + 0.7 * torch.rand(1, generator=g)))
    return (m * (0.5 + 0.5 * torch.rand(B, H, W, generator=g)) + 0.2 * torch.rand(B, H, W, generator=g) ** 4).clamp(0, 1)


def test_annotation_size_matches_oracle():
    """384x1280 (the annotation resolution), a long spiral that crosses hundreds of tiles, batch of 2."""
    from mindtheedge_amd.utils import tools
    p = _edge_map(2, 384, 1280, seed=4)
    nms = tools.non_max_suppression(p.cuda())
    for b in range(2):
        same(nms[b], do.non_max_suppression(p[b].numpy()))
        np.testing.assert_array_equal(tools.sobel_normals(p[b].cuda()).cpu().numpy(), do.normals_u8(p[b].numpy()))
    hy = tools.hysteresis(nms)
    for b in range(2):
        same(hy[b], do.hysteresis(nms[b].double().cpu().numpy()))
    assert 0 < int((hy > 0).sum()) < int((nms > 0).sum())
    # idempotence: survivors are exactly the strong components, a second pass keeps them all
    same(tools.hysteresis(hy), hy.double().cpu().numpy())
    # a weak spiral fed from one strong pixel
    H, W = 384, 1280
    s = np.zeros((H, W), np.float32)
    for k, row in enumerate(range(4, H - 4, 6)):
        s[row, 4:W - 4] = 0.5
        s[row:row + 7, (W - 5) if k % 2 == 0 else 4] = 0.5
    s[4, 4] = 0.95
    s[H - 2, 10:40] = 0.5                                   # touches nothing strong
    got = tools.hysteresis(torch.from_numpy(s).cuda())
    same(got, do.hysteresis(s.astype(np.float64)), rtol=1e-7)
    assert int((got > 0).sum()) > 60 * 1200


def test_annotate_edges_chain_matches_oracle():
    from mindtheedge_amd.utils import tools
    preds = [(_edge_map(2, 384 >> s, 1280 >> s, seed=9 + s) * 2.0).unsqueeze(1).cuda() for s in range(4)]     # network output in [0,2]
    out = tools.annotate_edges(preds)
    assert len(out) == 4
    for s, (e, n) in enumerate(out):
        for b in range(2):
            we, wn = do.annotate((preds[s][b, 0].cpu().numpy() / 2).astype(np.float32))
            same(e[b], we)
            np.testing.assert_array_equal(n[b].cpu().numpy(), wn)
    e1, n1 = tools.annotate_edges(preds, nms=False, hysteresis_=False, normals=False, scales=1)[0]
    assert n1 is None and torch.equal(e1, preds[0][:, 0] * 0.5)


def test_degenerate_shapes_and_errors():
    from mindtheedge_amd.utils import tools
    from mindtheedge_amd.kernels import MteError
    for H, W in [(1, 1), (1, 7), (2, 2), (3, 3), (5, 2), (4, 70), (20, 65)]:
        g = torch.Generator().manual_seed(H * 100 + W)
        p = torch.rand(H, W, generator=g)
        same(tools.non_max_suppression(p.cuda()), do.non_max_suppression(p.numpy()))
        np.testing.assert_array_equal(tools.sobel_normals(p.cuda()).cpu().numpy(), do.normals_u8(p.numpy()))
        same(tools.hysteresis(p.cuda()), do.hysteresis(p.double().numpy()), rtol=1e-7)
    with pytest.raises(MteError):
        tools.hysteresis(torch.zeros(4, 4))
    with pytest.raises(ValueError):
        tools.non_max_suppression(torch.zeros(1, 1, 4, 4).cuda())


def test_annotate_frame_driver_core():
    """infer_edge_estimation.annotate_frame: RGB pass + RGB+LiDAR pass (SAN branch) + post-processing, all on device."""
    import infer_edge_estimation as iee
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    from mindtheedge_amd.utils.config import load_config
    K.set_compute_dtype("bf16")
    cfg = load_config(None, {"model": {"depth_net": {"with_san": True}, "loss": {"supervised_method": "sparse-silog", "supervised_num_scales": 1,
                                                                                  "supervised_loss_weight": 1.0, "edges_depth_edge_loss_all_scales": True}}})
    torch.manual_seed(2)
    wrap = ModelWrapper(cfg).cuda().eval()
    g = torch.Generator().manual_seed(4)
    image = torch.rand(1, 3, 64, 128, generator=g).cuda()
    lidar = ((torch.rand(1, 1, 64, 128, generator=g) < 0.1).float() * 40.0).cuda()
    out = iee.annotate_frame(wrap, image, lidar)
    assert set(out) == {"regular", "lidar"} and len(out["regular"]) == len(out["lidar"]) == 4
    seen = []                                                          # the network outputs the driver post-processed
    hook = wrap.model.depth_net.register_forward_hook(lambda m, a, r: seen.append([t.float().clone() for t in r["inv_depths"][0]]))
    out = iee.annotate_frame(wrap, image, lidar)
    hook.remove()
    assert len(seen) == 2
    for key, preds in (("regular", seen[0]), ("lidar", seen[1])):
        for s, (e, n) in enumerate(out[key]):
            assert e.shape == (1, 64 >> s, 128 >> s) and n.dtype == torch.uint8
            we, wn = do.annotate((preds[s][0, 0].cpu().numpy() / 2).astype(np.float32))
            same(e[0], we)
            np.testing.assert_array_equal(n[0].cpu().numpy(), wn)
    assert not torch.equal(seen[0][0], seen[1][0])                     # the LiDAR pass differs from the RGB pass
    single = iee.annotate_frame(wrap, image, None, multiscale=False, nms=False, hysteresis=False, normals=False)
    assert set(single) == {"regular"} and len(single["regular"]) == 1 and single["regular"][0][1] is None
