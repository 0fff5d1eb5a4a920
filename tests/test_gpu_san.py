"""GPU (-m gpu): the sparse auxiliary (SAN) branch, inference only against oracle/san_oracle.py.
PARITY UNPINNED -- both sides state MinkowskiEngine's published semantics in dense form; MinkowskiEngine itself is not
available (see the oracle header).  These tests establish that the HIP path and the independent torch statement agree."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import san_oracle as so

pytestmark = pytest.mark.gpu


def _lidar(B, H, W, seed, density=0.06):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, 1, H, W, generator=g) < density).float() * (2.0 + 70.0 * torch.rand(B, 1, H, W, generator=g))


def _randomise(enc, seed):
    g = torch.Generator().manual_seed(seed)
    for name, p in list(enc.named_parameters()) + list(enc.named_buffers()):
        if name.endswith("kernel"):
            p.data.copy_((torch.rand(p.shape, generator=g) - 0.5) * (3.0 / (p.shape[0] * p.shape[1]) ** 0.5))
        elif name.endswith("running_var"):
            p.data.copy_(0.5 + torch.rand(p.shape, generator=g))
        elif name.endswith("num_batches_tracked"):
            continue
        else:
            p.data.copy_(torch.rand(p.shape, generator=g) - 0.3)


def test_glue_kernels_against_dense_statement():
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.layers.minkowski_encoder import MinkowskiEncoder
    K.set_compute_dtype("fp32")
    enc = MinkowskiEncoder([32, 64]).cuda().eval()
    d = _lidar(2, 32, 64, seed=1)
    enc.prep(d.cuda())
    feat, mask = enc.d
    assert torch.equal(mask.cpu().bool(), (d > 0)[:, 0])
    assert torch.equal(feat.float().cpu()[:, 0:1], d) and float(feat.float().abs()[:, 1:].sum()) == 0.0
    # pooling on signed features: the maximum runs over ACTIVE cells only
    g = torch.Generator().manual_seed(2)
    f = (torch.rand(2, 8, 32, 64, generator=g) - 0.7) * (d > 0)
    pooled = K.new_act(2, 8, 16, 32, torch.float32)
    m2 = torch.empty(2, 16, 32, dtype=torch.uint8, device="cuda")
    fa = K.as_act(f.cuda(), torch.float32)
    K.lib.mte_sparse_maxpool3s2(*K._pl(fa), mask.data_ptr(), *K._pl(pooled), m2.data_ptr(), 2, 32, 64, 8, K._dt(fa), K._stream())
    wf, wm = so.max_pool(f, d > 0)
    assert torch.equal(m2.cpu().bool(), wm[:, 0]) and torch.equal(pooled.float().cpu(), wf)
    assert float(wf.min()) < 0.0                                  # the case that distinguishes it from zero-filled pooling


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-4), ("bf16", 3e-2)])
def test_encoder_levels_match_dense_statement(dtype, tol):
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.layers.minkowski_encoder import MinkowskiEncoder
    K.set_compute_dtype(dtype)
    enc = MinkowskiEncoder([32, 64, 128, 256, 512]).eval()
    _randomise(enc, seed=4)
    P = {"mconvs." + k: v.detach().clone() for k, v in enc.state_dict().items()}
    assert "mconvs.mconvs.0.layer3.0.kernel" in P and P["mconvs.mconvs.0.layer3.0.kernel"].shape == (25, 1, 64)
    assert "mconvs.mconvs.2.layer_final.0.bn.running_mean" in P
    enc = enc.cuda()
    d = _lidar(2, 64, 128, seed=3)
    want = so.san_features(P, d)
    enc.prep(d.cuda())
    for level in range(5):
        got = enc().float().cpu()
        assert got.shape == want[level].shape
        assert rel_err(got, want[level]) < tol, (level, rel_err(got, want[level]))
        assert float(got.abs().sum()) > 0
        off = want[level] == 0
        assert float(got[off & (want[level].abs().sum(1, keepdim=True) == 0).expand_as(off)].abs().sum()) == 0.0    # zero off the active set


def test_packnetsan_with_lidar_input_matches_composition():
    """eval forward with input_depth: skips 1..4 and the bottleneck become skip * w + sparse + b before the decoder."""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
    K.set_compute_dtype("fp32")
    torch.manual_seed(0)
    net = PackNetSAN01(dropout=None, version="1A", with_san=True).cuda().eval()
    _randomise(net.mconvs, seed=9)
    net.weight.data.copy_(torch.tensor([0.9, 1.1, 0.8, 1.2, 1.0]))
    net.bias.data.copy_(torch.tensor([0.01, -0.02, 0.03, 0.0, -0.01]))
    assert not any(p.requires_grad for p in net.mconvs.parameters())
    assert sum(p.numel() for p in net.parameters() if p.requires_grad) == 76997806          # the training parameter set is unchanged
    rgb = torch.rand(1, 3, 64, 128, generator=torch.Generator().manual_seed(1)).cuda()
    d = _lidar(1, 64, 128, seed=6).cuda()
    with torch.no_grad():
        plain = net(rgb)["inv_depths"]
        fused = net(rgb, input_depth=d)["inv_depths"]
    inv0, feats0 = plain
    inv1, feats1 = fused
    assert rel_err(inv1[0].float().cpu(), inv0[0].float().cpu()) > 1e-4                      # the LiDAR input changes the prediction
    P = {k: v.detach().cpu() for k, v in net.state_dict().items() if k.startswith("mconvs.")}
    sparse = so.san_features(P, d.cpu())
    for level in range(5):
        want = feats0[level + 1].float().cpu() * float(net.weight[level].detach()) + sparse[level] + float(net.bias[level].detach())
        assert rel_err(feats1[level + 1].float().cpu(), want) < 3e-4, level
    assert rel_err(feats1[0].float().cpu(), feats0[0].float().cpu()) < 1e-5                  # full-resolution skip untouched
    with pytest.raises(NotImplementedError):
        PackNetSAN01(dropout=None, version="1A").cuda().eval()(rgb, input_depth=d)
    net.train()
    with pytest.raises(NotImplementedError):
        net(rgb, input_depth=d)


def test_validation_uses_the_lidar_pass_when_the_branch_exists():
    """Reference behaviour: in eval mode SemiSupEdgeModel forwards batch['input_depth'] (SemiSupEdgeModel.py:44), so
    evaluate_depth validates the RGB+LiDAR prediction.  Without the branch (default build) the key is ignored."""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    from mindtheedge_amd.utils.config import load_config
    from mindtheedge_amd.utils.synthetic import synthetic_batch
    K.set_compute_dtype("fp32")
    base = {"model": {"loss": {"supervised_method": "sparse-silog", "supervised_num_scales": 1, "supervised_loss_weight": 1.0,
                               "edges_depth_edge_loss_all_scales": True}}}
    batch = synthetic_batch(1, 64, 128, seed=3, device=torch.device("cuda", 0))
    batch.pop("edge")
    batch["input_depth"] = _lidar(1, 64, 128, seed=8).cuda()
    outs = {}
    for with_san in (False, True):
        cfg = load_config(None, {**base, "model": {**base["model"], "depth_net": {"with_san": with_san}}})
        torch.manual_seed(5)
        wrap = ModelWrapper(cfg).cuda().eval()
        if with_san:
            _randomise(wrap.depth_net.mconvs, seed=2)
        with torch.no_grad():
            outs[with_san] = wrap.model(dict(batch))["inv_depths"][0][0].float().cpu()
            if with_san:
                direct = wrap.depth_net(batch["rgb"], input_depth=batch["input_depth"])["inv_depths"][0][0].float().cpu()
                assert rel_err(outs[True], direct) < 1e-4
                m = wrap.evaluate_depth(dict(batch))["metrics"]
                assert set(m) == {"depth", "depth_pp", "depth_gt", "depth_pp_gt"}
    assert rel_err(outs[True], outs[False]) > 1e-4


def test_edge_estimation_lidar_model_eval():
    """registry name of the annotation config; eval forward = network(rgb, lidar/200) with the full-resolution map halved."""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    from mindtheedge_amd.utils.config import load_config
    K.set_compute_dtype("fp32")
    cfg = load_config(None, {"model": {"name": "EdgeEstimationLIDARModel", "loss": {"edges_depth_edge_loss_all_scales": True}}, "edges": {"train_depth_edges": False}})
    torch.manual_seed(7)
    wrap = ModelWrapper(cfg).cuda().eval()
    assert wrap.depth_net.with_san
    _randomise(wrap.depth_net.mconvs, seed=3)
    rgb = torch.rand(1, 3, 64, 128, generator=torch.Generator().manual_seed(2)).cuda()
    lidar = _lidar(1, 64, 128, seed=5).cuda()
    with torch.no_grad():
        out = wrap.model({"rgb": rgb, "input_depth": lidar.clone()})["inv_depths"][0]
        ref = wrap.depth_net(rgb, input_depth=lidar / 200.0)["inv_depths"][0]
    assert rel_err(out[0].float().cpu(), ref[0].float().cpu() / 2) < 1e-4
    assert rel_err(out[1].float().cpu(), ref[1].float().cpu()) < 1e-4          # only the full-resolution scale is halved in eval
    wrap.train()
    with pytest.raises(NotImplementedError):
        wrap.model({"rgb": rgb, "input_depth": lidar})
