"""GPU (-m gpu): validation depth metrics on device against the fixtures generated from the
reference (tests/golden/make_golden_metrics.py) and against the CPU oracle on seeded inputs, through the C ABI
(mte_depth_metrics / mte_post_process_inv_depth)."""
import glob
import os
import types

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as mo

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[len("metrics_"):-4] for p in glob.glob(os.path.join(GOLDEN, "metrics_*.npz"))
               if not p.endswith("post_process.npz"))
RTOL = 2e-5         # float32 logf/division on the device vs the reference's CPU float32; sums are float64 here


def _cfg(crop, scale_output, lo, hi):
    return types.SimpleNamespace(crop=crop, scale_output=scale_output, min_depth=lo, max_depth=hi)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("use_gt_scale", [False, True])
def test_depth_metrics_match_reference_fixture(name, use_gt_scale):
    from mindtheedge_amd.utils.depth import compute_depth_metrics
    z = np.load(os.path.join(GOLDEN, "metrics_%s.npz" % name))
    cfg = _cfg(str(z["crop"]), str(z["scale_output"]), float(z["min_depth"]), float(z["max_depth"]))
    got = compute_depth_metrics(cfg, torch.from_numpy(z["gt"]).cuda(), torch.from_numpy(z["pred"]).cuda(), use_gt_scale=use_gt_scale)
    assert got.is_cuda and got.dtype == torch.float32 and got.shape == (7,)
    np.testing.assert_allclose(got.cpu().numpy(), z["metrics_gt%d" % int(use_gt_scale)], rtol=RTOL, atol=1e-7)


@pytest.mark.parametrize("method", ["mean", "max", "min"])
def test_post_process_matches_reference_fixture(method):
    from mindtheedge_amd.utils.depth import post_process_inv_depth
    z = np.load(os.path.join(GOLDEN, "metrics_post_process.npz"))
    got = post_process_inv_depth(torch.from_numpy(z["inv_depth"]).cuda(), torch.from_numpy(z["inv_depth_flipped"]).cuda(), method)
    np.testing.assert_allclose(got.cpu().numpy(), z["pp_" + method], rtol=1e-6, atol=1e-7)


def _kitti_like(B, H, W, h, w, seed, holes=0.8):
    g = torch.Generator().manual_seed(seed)
    gt = 0.5 + 95.0 * torch.rand(B, 1, H, W, generator=g) ** 2
    gt = gt * (torch.rand(B, 1, H, W, generator=g) > holes).float()
    base = torch.nn.functional.interpolate(gt.clamp(min=1.0), size=(h, w), mode="nearest")
    pred = base * (0.6 + 0.8 * torch.rand(B, 1, h, w, generator=g))
    return gt, pred


@pytest.mark.parametrize("B,H,W,h,w,crop,scale", [(2, 375, 1242, 384, 1280, "garg", "resize"),     # KITTI ground truth vs network output
                                                   (1, 375, 1242, 352, 1216, "garg", "top-center"),
                                                   (3, 96, 320, 96, 320, "", "resize"),
                                                   (1, 7, 9, 3, 5, "", "resize"),
                                                   (2, 1, 33, 1, 17, "", "resize")])
def test_depth_metrics_match_oracle(B, H, W, h, w, crop, scale):
    from mindtheedge_amd.utils.depth import compute_depth_metrics
    gt, pred = _kitti_like(B, H, W, h, w, seed=H * 31 + w)
    for use_gt_scale in (False, True):
        want = mo.compute_depth_metrics(gt.numpy(), pred.numpy(), crop=crop, scale_output=scale, min_depth=0.0, max_depth=80.0,
                                        use_gt_scale=use_gt_scale)
        got = compute_depth_metrics(_cfg(crop, scale, 0.0, 80.0), gt.cuda(), pred.cuda(), use_gt_scale=use_gt_scale)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=RTOL, atol=1e-7)


def test_median_is_exact_rank_element():
    """With every pixel valid and pred == gt * c, median scaling must undo c exactly up to float32 rounding of the
    (p * med_g) / med_p expression; a1 = 1 and the error metrics ~ 0.  Also covers negative / tiny predictions."""
    from mindtheedge_amd.utils.depth import compute_depth_metrics
    g = torch.Generator().manual_seed(5)
    gt = 1.0 + 70.0 * torch.rand(2, 1, 40, 64, generator=g)
    pred = gt * 3.7
    m = compute_depth_metrics(_cfg("", "resize", 0.0, 80.0), gt.cuda(), pred.cuda(), use_gt_scale=True).cpu().numpy()
    assert m[4] == 1.0 and m[0] < 1e-6 and m[2] < 1e-4
    pred2 = pred.clone()
    pred2[:, :, ::2] *= -1.0                                   # half the predictions negative: the median is negative too
    want = mo.compute_depth_metrics(gt.numpy(), pred2.numpy(), use_gt_scale=True)
    got = compute_depth_metrics(_cfg("", "resize", 0.0, 80.0), gt.cuda(), pred2.cuda(), use_gt_scale=True).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-7)


def test_no_valid_pixel_anywhere_gives_zeros():
    from mindtheedge_amd.utils.depth import compute_depth_metrics
    gt = torch.zeros(2, 1, 16, 32)
    pred = torch.ones(2, 1, 16, 32)
    for s in (False, True):
        got = compute_depth_metrics(_cfg("garg", "resize", 0.0, 80.0), gt.cuda(), pred.cuda(), use_gt_scale=s)
        assert torch.equal(got.cpu(), torch.zeros(7))


def test_argument_errors_are_loud():
    from mindtheedge_amd.utils.depth import compute_depth_metrics, post_process_inv_depth
    from mindtheedge_amd.kernels import MteError
    x = torch.ones(1, 1, 4, 4)
    with pytest.raises(MteError):
        compute_depth_metrics(_cfg("", "resize", 0.0, 80.0), x, x)                      # CPU tensors: no fallback
    with pytest.raises(NotImplementedError):
        compute_depth_metrics(_cfg("", "bottom", 0.0, 80.0), x.cuda(), x.cuda())
    with pytest.raises(ValueError):
        post_process_inv_depth(x.cuda(), x.cuda(), "median")


def test_evaluate_depth_matches_oracle_on_network_output():
    """ModelWrapper.evaluate_depth: two network passes + fusion + 4 metric modes on device vs the oracle applied to the
    same network outputs."""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    from mindtheedge_amd.utils.config import load_config
    from mindtheedge_amd.utils.synthetic import synthetic_batch
    K.set_compute_dtype("bf16")
    cfg = load_config(None, {"model": {"loss": {"supervised_method": "sparse-silog", "supervised_num_scales": 1, "supervised_loss_weight": 1.0,
                                                  "edges_depth_edge_loss_all_scales": True},
                                       "depth_net": {"dropout": 0.5}, "params": {"crop": "garg"}}})
    torch.manual_seed(3)
    wrap = ModelWrapper(cfg).cuda().eval()
    batch = synthetic_batch(2, 64, 128, seed=11, device=torch.device("cuda", 0))
    gt, _ = _kitti_like(2, 61, 120, 64, 128, seed=2, holes=0.5)
    batch["depth"] = gt.cuda()
    rgb0 = batch["rgb"].clone()
    seen = []                                                                           # (input rgb, full-resolution inverse depth) per pass
    hook = wrap.model.register_forward_hook(lambda mod, args, res: seen.append((args[0]["rgb"].clone(), res["inv_depths"][0][0][:, 0:1].float().clone())))
    out = wrap.evaluate_depth(batch)
    hook.remove()
    assert torch.equal(batch["rgb"], rgb0)                                              # the caller's batch is not flipped
    assert len(seen) == 2 and torch.equal(seen[0][0], rgb0) and torch.equal(seen[1][0], torch.flip(rgb0, [3]))
    inv, invf = seen[0][1], seen[1][1]
    pp = mo.post_process_inv_depth(inv.cpu().numpy(), invf.cpu().numpy(), "mean")
    np.testing.assert_allclose(out["inv_depth"].cpu().numpy(), pp, rtol=1e-6, atol=1e-7)
    depth = (1.0 / inv.clamp(min=1e-6)).cpu().numpy()
    depth_pp = 1.0 / np.maximum(pp, np.float32(1e-6))
    for mode in ("", "_pp", "_gt", "_pp_gt"):
        want = mo.compute_depth_metrics(gt.numpy(), depth_pp if "pp" in mode else depth, crop="garg", use_gt_scale="gt" in mode)
        np.testing.assert_allclose(out["metrics"]["depth" + mode].cpu().numpy(), want, rtol=5e-5, atol=1e-7)
    # edge metrics of the first image: Canny (parity-unpinned restatement) x 3 settings -> chamfer both ways
    from oracle import canny_oracle as co
    from oracle import edge_oracle as eo
    gt_edge = batch["edge"][0, 0].cpu().numpy() * 255
    want_e = []
    for e in co.edges_from_depth(depth[0, 0]):
        want_e.extend(eo.precision_recall_f1(e, gt_edge))
    np.testing.assert_allclose(out["metrics"]["edges"].cpu().numpy(), np.array(want_e, np.float64), rtol=1e-12, equal_nan=True)
    summary = wrap.validation_epoch_end([{"idx": None, **out["metrics"]}, {"idx": None, **out["metrics"]}])
    assert abs(summary["depth-abs_rel_pp_gt"] - float(out["metrics"]["depth_pp_gt"][0])) < 1e-6
    assert len(summary) == 28 + 9 and "edges-f1_2" in summary


def test_trainer_validate_loop():
    """Trainer.validate over two synthetic validation sets: per-batch metrics stay on the device, one dict per dataset."""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    from mindtheedge_amd.trainers.trainer import Trainer
    from mindtheedge_amd.utils.config import load_config
    from mindtheedge_amd.utils.synthetic import synthetic_batch
    K.set_compute_dtype("bf16")
    cfg = load_config(None, {"model": {"loss": {"supervised_method": "sparse-silog", "supervised_num_scales": 1, "supervised_loss_weight": 1.0,
                                                  "edges_depth_edge_loss_all_scales": True}, "params": {"crop": "garg"}}})
    torch.manual_seed(3)
    wrap = ModelWrapper(cfg).cuda()
    wrap.train()

    def loader(seed, n):
        out = []
        for i in range(n):
            b = synthetic_batch(2, 64, 128, seed=seed + i, device=torch.device("cpu"))
            b["depth"] = _kitti_like(2, 61, 120, 64, 128, seed=seed + i, holes=0.5)[0]
            b.pop("edge", None)
            out.append(b)
        return out
    res = Trainer(max_epochs=1).validate([loader(1, 2), loader(9, 1)], wrap)
    assert wrap.training and len(res) == 2
    for r in res:
        assert len(r) == 28 and all(np.isfinite(v) for v in r.values())
        assert 0.0 <= r["depth-a1_pp_gt"] <= r["depth-a2_pp_gt"] <= r["depth-a3_pp_gt"] <= 1.0
