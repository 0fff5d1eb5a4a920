"""GPU (-m gpu): the one-launch loss path (kernels.DepthLossesFn -> mte_edge_loss_multi_fwd / _bwd: depth-edge loss of all four
scales + silog of scale 0, finalised by the last workgroup) against the CPU oracle (oracle/loss_oracle.py, pinned by
tests/golden/loss_*.npz and model_semisup_64x128.npz) -- values AND gradients with respect to all four inverse-depth maps:

* ragged sizes: widths that are not a multiple of 4 (scalar access path), of 64 (partial tiles) and heights that are not a
  multiple of 32; a single tile; many tiles;
* with / without normals (direction-selected |Sobel| vs magnitude), with a binary mask, an all-ones mask, without mask;
* the all-negative batch (alpha = 1 branch), inverse depths at and below the 1e-6 clamp;
* the fused path against the per-scale head calls of the same model (identical composition);
* BASELINE size 384x1280, B = 2, and run-to-run bit-identical sums (fixed-order reduction, no atomics)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _maps(B, H, W, seed, with_normals=True, mask=None, all_negative=False, clamp_cases=False):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    invs, batch = [], {}
    for s in range(4):
        h, w = H >> s, W >> s
        inv = 0.05 + 1.9 * r(B, 1, h, w)                    # sigmoid/0.5 range (0, 2): depth up to 20 m -> strong Sobel responses
        if clamp_cases and s == 0:
            inv.view(-1)[:7] = torch.tensor([1e-6, 5e-7, 0.0, -1.0, 1.0000001e-6, 9.99e-7, 2e-6])
        invs.append(inv)
        sfx = "" if s == 0 else "_%d" % s
        on = (r(B, 1, h, w) < (0.0 if all_negative else 0.08)).float()
        batch["edge" + sfx] = on * r(B, 1, h, w)
        if with_normals:
            batch["normal" + sfx] = (r(B, 1, h, w) * 2 - 1) * math.pi
    batch["depth"] = (r(B, 1, H, W) < 0.1).float() * (1.0 + 79.0 * r(B, 1, H, W))
    if mask == "binary":
        batch["rgb_edge"] = (r(B, 1, H, W) < 0.7).float()
    elif mask == "ones":
        batch["rgb_edge"] = torch.ones(B, 1, H, W)
    return invs, batch


def _oracle(invs, batch, mask_all_scales):
    from oracle import loss_oracle as lo
    invs = [i.double().requires_grad_(True) for i in invs]
    b = {k: v.double() for k, v in batch.items()}
    if mask_all_scales is False:
        b.pop("rgb_edge", None)
    out = lo.semisup_edge_model_loss(invs, b)
    out["loss"].sum().backward()
    return float(out["loss"]), float(out["edge_loss"]), float(out["supervised_loss"]), [i.grad for i in invs]


def _model(fuse):
    from mindtheedge_amd.models.SemiSupEdgeModel import SemiSupEdgeModel
    from mindtheedge_amd.losses.grad_loss import GradLoss
    m = SemiSupEdgeModel(supervised_loss_weight=1.0, depth_edges_loss_weight=1.0, supervised_method="sparse-silog",
                         supervised_num_scales=1, edges_depth_edge_loss_all_scales=True, flip_lr_prob=0.0)
    m.add_edge_loss(GradLoss("cross_entropy", True, [], 10.0, 1.0))
    m.fuse_losses = fuse
    return m


def _hip(invs, batch, fuse=True):
    m = _model(fuse)
    dev = torch.device("cuda")
    invs = [i.to(dev).requires_grad_(True) for i in invs]
    b = {k: v.to(dev) for k, v in batch.items()}
    if fuse:
        edge, sup = m._fused_losses(invs, b)
    else:
        edge = m.compute_edge_loss_with_all_scales(invs, b, b.get("rgb_edge"), is_grad=True, is_sigmoid=True, sigmoid_thresh=4)
        sup = m.supervised_loss(invs, b["depth"])["loss"]
    loss = sup + edge
    loss.sum().backward()
    torch.cuda.synchronize()
    return float(loss.sum()), float(edge), float(sup.sum()), [i.grad.cpu() for i in invs]


def _close(got, want, tol, gtol):
    assert got[0] == pytest.approx(want[0], rel=tol)
    assert got[1] == pytest.approx(want[1], rel=tol) and got[2] == pytest.approx(want[2], rel=tol)
    for s, (a, b) in enumerate(zip(got[3], want[3])):
        err = float((a.double() - b).abs().max() / b.abs().max().clamp(min=1e-30))
        assert err <= gtol, (s, err)


@pytest.mark.parametrize("shape", [(2, 64, 128), (1, 32, 64), (3, 96, 200), (2, 72, 100), (1, 40, 36), (2, 128, 320)])
@pytest.mark.parametrize("normals", [True, False])
def test_fused_losses_match_the_oracle_without_mask(shape, normals):
    B, H, W = shape
    invs, batch = _maps(B, H, W, seed=H + W + B, with_normals=normals)
    want = _oracle(invs, batch, None)
    got = _hip(invs, batch, fuse=True)
    _close(got, want, 1e-4, 2e-4)
    per_scale = _hip(invs, batch, fuse=False)
    _close(per_scale, want, 1e-4, 2e-4)
    _close(got, (per_scale[0], per_scale[1], per_scale[2], [g.double() for g in per_scale[3]]), 2e-6, 2e-5)


def test_all_negative_batch_and_clamped_inverse_depths():
    invs, batch = _maps(2, 64, 128, seed=5, all_negative=True, clamp_cases=True)
    invs[0] = invs[0].clamp(min=-1.0)
    want = _oracle(invs, batch, None)
    got = _hip(invs, batch)
    # no positive label anywhere: alpha = 1 for every sample, the negative term gets weight 1 - alpha = 0 -> edge loss exactly 0
    assert want[1] == 0.0 and got[1] == 0.0
    for a, b in zip(got[3][1:], want[3][1:]):                # ... and so are its gradients (scales 1-3 carry the edge loss only)
        assert float(b.abs().max()) == 0.0 and float(a.abs().max()) == 0.0
    # scale 0 carries the silog gradient; inv <= 0 makes the reference's log NaN / inf as well: compare where it is finite
    g0, w0 = got[3][0].double().flatten(), want[3][0].flatten()
    ok = torch.isfinite(w0)
    assert int(ok.sum()) > 0.9 * ok.numel()
    assert float((g0[ok] - w0[ok]).abs().max() / w0[ok].abs().max()) <= 2e-4


def test_clamped_inverse_depths_with_edges():
    """inv at / below the 1e-6 clamp of inv2depth: depth = 1e6 there, gradient passes only where inv >= 1e-6 (torch.clamp)."""
    from oracle import loss_oracle as lo
    from mindtheedge_amd.losses.grad_loss import GradLoss
    invs, batch = _maps(1, 32, 64, seed=3, clamp_cases=True)
    # float32 like the reference: whether inv == float32(1e-6) counts as clamped is decided in float32 there (it is not)
    inv = invs[0].clone().requires_grad_(True)
    b = batch
    want, _ = lo.grad_loss(lo.inv2depth(inv), b["edge"], None, True, True, 4.0, b["normal"])
    want.backward()
    x = invs[0].cuda().requires_grad_(True)
    got, _ = GradLoss("cross_entropy", True, [], 10.0, 1.0)(x, batch["edge"].cuda(), None, True, True, 4, batch["normal"].cuda(), from_inv_depth=True)
    got.backward()
    assert float(got) == pytest.approx(float(want), rel=1e-4)
    gw = inv.grad.double().flatten()
    gg = x.grad.cpu().double().flatten()
    assert all(float(gg[i]) == 0.0 and float(gw[i]) == 0.0 for i in (1, 2, 3, 5))   # inv < 1e-6: clamped, no gradient
    assert float(gg[0]) != 0.0 and float(gw[0]) != 0.0                             # inv == 1e-6: gradient passes (x >= min)
    # depth = 1e6 next to depth ~ 1: the Sobel sums cancel catastrophically in float32 on both sides; compare the well-conditioned part
    far = torch.ones(32, 64, dtype=torch.bool)
    far[:3, :12] = False
    far = far.flatten()
    assert float((gg[far] - gw[far]).abs().max()) <= 1e-3 * float(gw[far].abs().max())


@pytest.mark.parametrize("mask", ["binary", "ones"])
def test_single_scale_head_with_mask_through_the_new_kernels(mask):
    """The reference passes ONE full-resolution mask to every scale (only valid at scale 0): the masked branches are checked on
    the single-scale entry points (GradLoss called directly), which run on the same kernels."""
    from oracle import loss_oracle as lo
    from mindtheedge_amd.losses.grad_loss import GradLoss
    invs, batch = _maps(2, 72, 100, seed=9, mask=mask)
    inv = invs[0].double().requires_grad_(True)
    b = {k: v.double() for k, v in batch.items()}
    want, _ = lo.grad_loss(lo.inv2depth(inv), b["edge"], b["rgb_edge"], True, True, 4.0, b["normal"])
    want.backward()
    head = GradLoss("cross_entropy", True, [], 10.0, 1.0)
    x = invs[0].cuda().requires_grad_(True)
    got, _ = head(x, batch["edge"].cuda(), batch["rgb_edge"].cuda(), True, True, 4, batch["normal"].cuda(), from_inv_depth=True)
    got.backward()
    assert float(got) == pytest.approx(float(want), rel=1e-4)
    assert float((x.grad.cpu().double() - inv.grad).abs().max() / inv.grad.abs().max()) <= 2e-4


def test_full_size_384x1280_and_determinism():
    invs, batch = _maps(2, 384, 1280, seed=1)
    want = _oracle(invs, batch, None)
    got = _hip(invs, batch)
    for a, b in zip(got[:3], want[:3]):
        assert a == pytest.approx(b, rel=1e-4)
    # |Sobel| is not differentiable at 0: at the one-in-a-million pixel whose directional response cancels to ~1e-6 the float32
    # arithmetic of the reference (and of this kernel) and the float64 oracle pick opposite signs, and the 8 neighbours of that
    # pixel receive +-dg instead of -+dg.  Everything else must agree to 2e-4 of the largest gradient.
    for s_, (a, b) in enumerate(zip(got[3], want[3])):
        d = (a.double() - b).abs() / b.abs().max()
        outliers = int((d > 2e-4).sum())
        assert outliers <= 18, (s_, outliers, float(d.max()))                 # at most two such pixels (9 taps each) per scale
        assert float(d.max()) <= 2e-2, (s_, float(d.max()))
        ok = d <= 2e-4
        assert float((a.double() - b)[ok].pow(2).mean().sqrt() / b[ok].pow(2).mean().sqrt()) <= 1e-4
    # round 4: no floating-point atomics on the loss path any more -- one record per workgroup, image sums and loss sums added in a fixed
    # order (forward), no accumulation at all in the backward: losses AND gradients are bit-reproducible, also with other work on the chip
    noise = torch.cuda.Stream()
    na = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    nb = torch.empty_like(na)
    for rep in range(3):
        if rep:
            with torch.cuda.stream(noise):
                nb.copy_(na)
        again = _hip(invs, batch)
        assert again[:3] == got[:3]
        for a, b in zip(got[3], again[3]):
            assert torch.equal(a, b)
    torch.cuda.synchronize()


def test_library_clears_its_own_workspace_unless_told_otherwise():
    """MTE_OPT_LOSS_PREZEROED off (the C-ABI default a hand-written binding gets): `work` may hold anything -- the library fills what it needs.
    On: the caller's zeros are trusted (kernels.py carves the workspace from its zeroed arena)."""
    import ctypes
    from mindtheedge_amd import kernels as K
    invs, batch = _maps(2, 96, 160, seed=9)
    dev = torch.device("cuda")
    preds = [i.to(dev).contiguous() for i in invs]
    sfx = ["", "_1", "_2", "_3"]
    edges = [batch["edge" + s].to(dev).contiguous() for s in sfx]
    normals = [batch["normal" + s].to(dev).contiguous() for s in sfx]
    gt = batch["depth"].to(dev).contiguous()
    arr = (K._EdgeScale * 4)()
    for o, p, e, n in zip(arr, preds, edges, normals):
        o.pred, o.edge, o.normal, o.mask, o.gmap, o.dpred = p.data_ptr(), e.data_ptr(), n.data_ptr(), None, None, None
        o.H, o.W = p.shape[-2], p.shape[-1]
    n = K.lib.mte_edge_loss_work_elems(ctypes.addressof(arr), 4, 2)
    results = []
    try:
        for prezeroed, fill in ((0, 123.0), (0, float("nan")), (1, 0.0)):
            K.lib.set_option(1, prezeroed)
            work = torch.full((n,), fill, dtype=torch.float64, device=dev)
            losses = torch.empty((5,), dtype=torch.float32, device=dev)
            coef = torch.empty((4 * 5,), dtype=torch.float32, device=dev)
            aux = torch.empty((2,), dtype=torch.float32, device=dev)
            K.lib.mte_edge_loss_multi_fwd(ctypes.addressof(arr), 4, 2, 1, 1, 1, 4.0, 10.0, 1.0, gt.data_ptr(), work.data_ptr(), losses.data_ptr(),
                                          coef.data_ptr(), losses.data_ptr() + 16, aux.data_ptr(), K._stream())
            torch.cuda.synchronize()
            results.append((losses.clone(), coef.clone(), aux.clone()))
    finally:
        K.lib.set_option(1, 1 if K._arena.enabled else 0)
    for r in results[1:]:
        for a, b in zip(results[0], r):
            assert torch.equal(a, b)
    assert bool(torch.isfinite(results[0][0]).all())
