"""GPU (-m gpu): the sparse auxiliary (SAN) branch, inference and training against oracle/san_oracle.py.
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


@pytest.mark.parametrize("density", [0.0, 0.05, 0.5, 1.0])
def test_site_list_is_the_raster_ordered_active_set(density):
    """mte_sparse_site_list: three small launches, no atomics -> exactly torch.nonzero's order; the count stays on the device"""
    from mindtheedge_amd import kernels as K
    g = torch.Generator().manual_seed(int(density * 100))
    for shape in ((2, 24, 40), (1, 192, 640), (3, 7, 13)):
        mask = (torch.rand(shape, generator=g) < density).to(torch.uint8).cuda()
        sl = K.SiteList(mask)
        torch.cuda.synchronize()
        n = int(sl.count[0])
        want = torch.nonzero(mask.flatten()).flatten().int()
        assert n == want.numel()
        assert torch.equal(sl.rows[:n], want)


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
@pytest.mark.parametrize("shape,density", [((64, 64, 3, 2, 24, 40), 0.05), ((8, 64, 5, 1, 48, 64), 0.06), ((128, 256, 3, 2, 12, 20), 0.5),
                                           ((32, 32, 3, 1, 16, 32), 1.0), ((64, 128, 3, 1, 16, 32), 0.0)])
def test_sparse_convolution_equals_the_dense_form_on_the_active_sites(shape, density, dtype):
    """gather-GEMM-scatter over the site list (mte_conv2d_igemm_sparse) against the dense convolution of the same zero-filled map:
    the same products in the same order per output, so forward and data gradient are BIT-identical on the active sites, and
    nothing is written off them"""
    from mindtheedge_amd import kernels as K
    cin, cout, k, B, H, W = shape
    K.set_compute_dtype(dtype)
    try:
        g = torch.Generator().manual_seed(cin + cout + k)
        mask = (torch.rand(B, H, W, generator=g) < density)
        x = (torch.rand(B, cin, H, W, generator=g) * 2 - 1) * mask[:, None]
        dy = (torch.rand(B, cout, H, W, generator=g) * 2 - 1) * mask[:, None]
        w = ((torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * (3.0 / (cin * k * k)) ** 0.5).cuda()
        xa, dya = K.as_act(x.cuda(), K.compute_dtype()), K.as_act(dy.cuda(), K.compute_dtype())
        sites = K.SiteList(mask.to(torch.uint8).cuda())
        pack = K.WeightPack()
        wf, wb = pack.get(w, xa.dtype, True)
        K.use_patch_kernels(False)                                    # (the dense side on the implicit GEMM too, un-split: same accumulation order)
        orig, K._splitk_workspace = K._splitk_workspace, lambda *a: (None, 0)
        try:
            dense = K.conv_forward(xa, wf, None, cout, k, k).float()
            ddx, _, _ = K.conv_backward(xa, dya, w, pack, True, need_dw=False)
        finally:
            K._splitk_workspace = orig
            K.use_patch_kernels(True)
        out = torch.full_like(K.new_act(B, cout, H, W), 7.0)
        got = K.conv_forward(xa, wf, None, cout, k, k, out=out, sites=sites).float()
        sdx, _, _ = K.conv_backward(xa, dya, w, pack, True, need_dw=False, sites=sites)
        torch.cuda.synchronize()
        m = mask.cuda()[:, None]
        assert torch.equal(torch.where(m, got, torch.zeros_like(got)), torch.where(m, dense, torch.zeros_like(dense)))
        assert bool((got[(~m).expand_as(got)] == 7.0).all())         # untouched off the active set
        mi = m.expand(B, K.round8(cin), H, W)
        assert torch.equal(torch.where(mi, sdx.float(), torch.zeros_like(sdx.float())), torch.where(mi, ddx.float(), torch.zeros_like(ddx.float())))
        if density > 0:
            assert float(dense.abs().max()) > 0
    finally:
        K.use_patch_kernels(True)
        K.set_compute_dtype("bf16")


@pytest.mark.parametrize("gather", [True, False])
@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-4), ("bf16", 3e-2)])
def test_encoder_levels_match_dense_statement(dtype, tol, gather, monkeypatch):
    """gather = True: the convolutions run over the active sites (round 3); False: the dense-equivalent form of round 2"""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.layers import minkowski_encoder as me
    from mindtheedge_amd.networks.layers.minkowski_encoder import MinkowskiEncoder
    monkeypatch.setattr(me, "SPARSE_GATHER", gather)
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
        got = enc().detach().float().cpu()
        assert got.shape == want[level].shape
        assert rel_err(got, want[level]) < tol, (level, rel_err(got, want[level]))
        assert float(got.detach().abs().sum()) > 0
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
    assert sum(p.numel() for p in PackNetSAN01(dropout=None, version="1A").parameters() if p.requires_grad) == 76997806   # default build: dense set only
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


# ---------------------------------------------------------------------------------------------------------------------
# training path (round 2): backward kernels of the branch against torch autograd through the dense statement
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_params(enc):
    P = {}
    for k, v in enc.state_dict().items():
        t = v.detach().cpu().clone()
        if t.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            t.requires_grad_(True)
        P["mconvs." + k] = t
    return P


@pytest.mark.parametrize("gather", [True, False])
def test_training_features_gradients_and_running_statistics_match_autograd(gather, monkeypatch):
    from mindtheedge_amd.networks.layers import minkowski_encoder as me
    monkeypatch.setattr(me, "SPARSE_GATHER", gather)
    _training_features_gradients_and_running_statistics()


def _training_features_gradients_and_running_statistics():
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.layers.minkowski_encoder import MinkowskiEncoder
    K.set_compute_dtype("fp32")
    enc = MinkowskiEncoder([32, 64, 128, 256, 512])
    _randomise(enc, seed=11)
    P = _oracle_params(enc)
    enc = enc.cuda().train()
    d = _lidar(2, 64, 128, seed=12, density=0.12)
    g = torch.Generator().manual_seed(13)
    want = so.san_features(P, d, train=True)
    R = [torch.rand(w.shape, generator=g) - 0.5 for w in want]
    sum((w * r).sum() for w, r in zip(want, R)).backward()
    enc.prep(d.cuda())
    got = [enc() for _ in range(5)]
    for level in range(5):
        assert rel_err(got[level].float().cpu(), want[level].detach()) < 3e-4, level
    sum((K_.float() * r.cuda()).sum() for K_, r in zip(got, R)).backward()
    torch.cuda.synchronize()
    checked = 0
    for name, p in enc.named_parameters():
        ref = P["mconvs." + name].grad
        assert p.grad is not None and ref is not None, name
        assert rel_err(p.grad.cpu(), ref) < 2e-3, (name, rel_err(p.grad.cpu(), ref))
        assert float(ref.abs().max()) > 0
        checked += 1
    assert checked == 70                                              # the reference's 70 mconvs parameter tensors
    for name, b in enc.named_buffers():                               # running statistics: momentum 0.1, unbiased variance
        if name.endswith(("running_mean", "running_var")):
            assert rel_err(b.cpu(), P["mconvs." + name]) < 1e-4, name
        elif name.endswith("num_batches_tracked"):
            assert int(b) == 1


def test_pooling_and_fusion_backward_kernels():
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.layers.minkowski_encoder import _SparseMaxPoolFn, san_fuse, feature_l2
    K.set_compute_dtype("fp32")
    g = torch.Generator().manual_seed(21)
    d = _lidar(2, 32, 64, seed=22, density=0.3)
    mask = (d > 0)
    f = ((torch.rand(2, 8, 32, 64, generator=g) - 0.6) * mask).requires_grad_(True)
    # ties on purpose: a quarter of the active values are set to one common value
    with torch.no_grad():
        f[(torch.rand(f.shape, generator=g) < 0.25) & mask.expand_as(f)] = 0.125
    want, _ = so.max_pool(f, mask)
    Rp = torch.rand(want.shape, generator=g) - 0.5
    (want * Rp).sum().backward()
    fa = K.as_act(f.detach().cuda(), torch.float32).requires_grad_(True)
    m8 = mask[:, 0].to(torch.uint8).cuda().contiguous()
    got, m2 = _SparseMaxPoolFn.apply(fa, m8)
    assert torch.equal(got.float().cpu(), want.detach())
    (got.float() * Rp.cuda()).sum().backward()
    assert torch.equal(fa.grad.float().cpu(), f.grad), float((fa.grad.float().cpu() - f.grad).abs().max())
    # fusion: skip * w[i] + sparse + b[i]
    skip = (torch.rand(2, 32, 16, 32, generator=g) - 0.5).requires_grad_(True)
    sp = (torch.rand(2, 32, 16, 32, generator=g) - 0.5).requires_grad_(True)
    w = torch.tensor([0.9, 1.1, 0.8, 1.2, 1.0], requires_grad=True)
    b = torch.tensor([0.01, -0.02, 0.03, 0.0, -0.01], requires_grad=True)
    Rf = torch.rand(2, 32, 16, 32, generator=g) - 0.5
    ((skip * w[2] + sp + b[2]) * Rf).sum().backward()
    sk = K.as_act(skip.detach().cuda(), torch.float32).requires_grad_(True)
    sq = K.as_act(sp.detach().cuda(), torch.float32).requires_grad_(True)
    wg, bg = w.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    out = san_fuse(sk, sq, wg, bg, 2)
    (out.float() * Rf.cuda()).sum().backward()
    assert rel_err(sk.grad.float().cpu(), skip.grad) < 1e-6 and rel_err(sq.grad.float().cpu(), sp.grad) < 1e-6
    assert rel_err(wg.grad.cpu(), w.grad) < 1e-5 and rel_err(bg.grad.cpu(), b.grad) < 1e-5
    assert float(wg.grad[0]) == 0.0 and float(wg.grad[2]) != 0.0
    # feature-matching loss: mean((a.detach() - b)^2), gradient into b only
    a_ = torch.rand(2, 32, 16, 32, generator=g)
    b_ = torch.rand(2, 32, 16, 32, generator=g).requires_grad_(True)
    (3.0 * ((a_ - b_) ** 2).mean()).backward()
    ag = K.as_act(a_.cuda(), torch.float32).requires_grad_(True)
    bg_ = K.as_act(b_.detach().cuda(), torch.float32).requires_grad_(True)
    loss = feature_l2(ag, bg_)
    assert abs(float(loss) - float(((a_ - b_) ** 2).mean())) < 1e-6
    (3.0 * loss).backward()
    assert ag.grad is None and rel_err(bg_.grad.float().cpu(), b_.grad) < 1e-5


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_packnetsan_two_pass_training_step(dtype):
    """train-mode forward with input_depth (reference PackNetSAN01.py:324-342): RGB pass, RGB+LiDAR pass, feature-matching loss;
    one backward reaches the dense network, the sparse branch and the fusion scalars"""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
    K.set_compute_dtype(dtype)
    torch.manual_seed(0)
    net = PackNetSAN01(dropout=None, version="1A", with_san=True).cuda().train()
    _randomise(net.mconvs, seed=9)
    assert sum(1 for _ in net.mconvs.parameters()) == 70 and all(p.requires_grad for p in net.mconvs.parameters())
    rgb = torch.rand(2, 3, 64, 128, generator=torch.Generator().manual_seed(1)).cuda()
    d = _lidar(2, 64, 128, seed=6, density=0.1).cuda()
    out = net(rgb, input_depth=d, output_features=True)
    assert set(out) == {"inv_depths", "inv_depths_rgbd", "depth_loss", "skip_feat_rgb", "skip_feat_rgbd"}
    assert len(out["inv_depths"]) == 4 and len(out["inv_depths_rgbd"]) == 4 and len(out["skip_feat_rgbd"]) == 6
    want = sum(((a.detach().float() - b.detach().float()) ** 2).mean() for a, b in zip(out["skip_feat_rgbd"], out["skip_feat_rgb"])) / 6
    assert abs(float(out["depth_loss"]) - float(want)) <= 2e-3 * float(want) + 1e-7
    assert float(out["depth_loss"]) > 0
    assert rel_err(out["inv_depths_rgbd"][0].float(), out["inv_depths"][0].float()) > 1e-4
    loss = out["depth_loss"] + sum(i.float().mean() for i in out["inv_depths"]) + sum(i.float().mean() for i in out["inv_depths_rgbd"])
    loss.backward()
    torch.cuda.synchronize()
    for name, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    assert all(float(p.grad.abs().max()) > 0 for p in net.mconvs.parameters())
    assert float(net.weight.grad.abs().min()) > 0 and float(net.bias.grad.abs().min()) > 0
    assert float(net.encoder.conv1.conv_base.weight.grad.abs().max()) > 0


def test_edge_estimation_lidar_model_training_step():
    """DEE training WITH a LiDAR input (reference EdgeEstimationLIDARModel.py:135-160):
    loss = depth_loss + (edge_rgb + weight_rgbd * edge_lidar) / 2, both edge terms BCE on the halved network output."""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    from mindtheedge_amd.utils.config import load_config
    from mindtheedge_amd.utils.synthetic import synthetic_batch
    K.set_compute_dtype("fp32")
    cfg = load_config(None, {"model": {"name": "EdgeEstimationLIDARModel", "loss": {"edges_depth_edge_loss_all_scales": True}},
                             "edges": {"train_depth_edges": True}})
    torch.manual_seed(7)
    wrap = ModelWrapper(cfg).cuda().train()
    assert wrap.depth_net.with_san
    _randomise(wrap.depth_net.mconvs, seed=3)
    batch = synthetic_batch(2, 64, 128, seed=4, device=torch.device("cuda", 0))
    batch["input_depth"] = _lidar(2, 64, 128, seed=5, density=0.1).cuda() * 2.0
    wrap.model._pinned_flip = False
    out = wrap.model(dict(batch))
    assert {"edge_loss", "edge_lidar_loss"} <= set(out["metrics"]) and "depth_loss" in out and "inv_depths_rgbd" in out
    head = wrap.model.edge_loss_head
    def edge_term(probs):
        tot = 0.0
        for s in range(4):
            l, _ = head(probs[s].detach(), batch["edge" if s == 0 else "edge_%d" % s], None, False, False, 0)
            tot = tot + float(l)
        return tot / 4
    e_rgb, e_lidar = edge_term(out["inv_depths"]), edge_term(out["inv_depths_rgbd"])
    assert abs(float(out["metrics"]["edge_loss"]) - e_rgb) < 1e-5 * max(1.0, abs(e_rgb))
    assert abs(float(out["metrics"]["edge_lidar_loss"]) - e_lidar) < 1e-5 * max(1.0, abs(e_lidar))
    want = float(out["depth_loss"]) + (e_rgb + wrap.model.weight_rgbd * e_lidar) / 2
    assert abs(float(out["loss"]) - want) < 1e-5 * max(1.0, abs(want))
    # the network saw input_depth / 200 (reference :108-110)
    with torch.no_grad():
        again = wrap.depth_net(batch["rgb"], input_depth=batch["input_depth"] / 200.0)
    assert rel_err(again["inv_depths_rgbd"][0].float() / 2, out["inv_depths_rgbd"][0].float()) < 1e-4
    out["loss"].backward()
    torch.cuda.synchronize()
    grads = [p.grad for p in wrap.depth_net.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    assert all(float(p.grad.abs().max()) > 0 for p in wrap.depth_net.mconvs.parameters())


def test_two_pass_gradients_with_the_flat_gradient_sink_equal_plain_autograd():
    """round-2 advisor finding: the RGB and the RGB+LiDAR pass share the 216 encoder / decoder tensors, and the flat gradient sink
    STORES a parameter's gradient (one store per zero_grad) -- the second pass used to overwrite the first.  With the sink suspended
    for shared-parameter graphs the gradients accumulated into the flat buffer must equal those of plain autograd accumulation, twice in
    a row (zero_grad re-arms the sink), and a bucket's all-reduce trigger must fire once per parameter, after its LAST use
    (reference PackNetSAN01.py:324-342: one backward through both passes)."""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
    from mindtheedge_amd.trainers.data_parallel import FlatParameters
    K.set_compute_dtype("fp32")
    K.set_grad_sink(None)
    try:
        rgb = torch.rand(2, 3, 64, 128, generator=torch.Generator().manual_seed(1)).cuda()
        d = _lidar(2, 64, 128, seed=6, density=0.1).cuda()

        def build():
            torch.manual_seed(0)
            net = PackNetSAN01(dropout=None, version="1A", with_san=True).cuda().train()
            _randomise(net.mconvs, seed=9)
            return net

        def loss_of(net):
            out = net(rgb, input_depth=d)
            return out["depth_loss"] + sum(i.float().mean() for i in out["inv_depths"]) + 2.0 * sum(i.float().mean() for i in out["inv_depths_rgbd"])

        ref_net = build()
        loss_of(ref_net).backward()
        ref = {n: p.grad.detach().clone() for n, p in ref_net.named_parameters()}
        net = build()
        flat = FlatParameters(net.parameters())
        assert flat.sink is not None
        fired = []
        flat.sink.on_ready = lambda p: fired.append(id(p))
        hooks = [p.register_post_accumulate_grad_hook(lambda p: fired.append(id(p))) for p in net.parameters()]
        for rep in range(2):
            flat.zero_grad()
            fired.clear()
            loss_of(net).backward()
            K.join_side_stream()
            torch.cuda.synchronize()
            assert flat.sink.suspended                                   # the two-pass forward asked for autograd accumulation
            worst = 0.0
            for n, p in net.named_parameters():
                e = float((p.grad - ref[n]).abs().max() / ref[n].abs().max().clamp(min=1e-30))
                worst = max(worst, e)
                assert e < 2e-4, (rep, n, e)                             # fp32 mode: atomics-order noise only
            assert len(fired) == len(set(fired)) == sum(1 for _ in net.parameters())     # once per parameter
        for h in hooks:
            h.remove()
        # a single-pass step afterwards goes back to in-place stores: the suspension ends when the optimizer consumes the gradients
        # (FusedAdam.step -> end_step) or at the second zero_grad after it -- it survives ONE zero_grad, so that forward -> zero_grad ->
        # backward keeps it (round-3 advisor finding)
        flat.zero_grad()
        assert flat.sink.suspended
        flat.sink.end_step()
        assert not flat.sink.suspended
        flat.sink.suspend(); flat.zero_grad(); flat.zero_grad()
        assert not flat.sink.suspended
        out = net(rgb)
        sum(i.float().mean() for i in out["inv_depths"]).backward()
        K.join_side_stream()
        torch.cuda.synchronize()
        assert not flat.sink.suspended and len(flat.sink.written) >= 200
    finally:
        K.set_grad_sink(None)
        K.set_compute_dtype("bf16")
