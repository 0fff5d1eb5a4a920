"""GPU (-m gpu): properties at BASELINE.json's full resolution (384x1280), where the oracle is too slow to be the checker.

* batch independence: GroupNorm is per sample and nothing else couples the frames of a batch, so frame i of a 3-frame
  batch must give the same inverse depths as frame i alone (different tile<->sample alignment, statistics atomics and
  kernel variants per launch, same arithmetic);
* the bf16 benchmark mode against the fp32 validation mode of the same kernels on a full training step (loss scalars to
  the north-star's 1e-3, gradient energy to bf16 tolerance);
* the LDS-patch conv family against the generic implicit GEMM on a full training step;
* the weight-gradient side stream against single-stream execution (the schedule must not change the result)."""
import os

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

H, W = 384, 1280


def _model(dtype, dropout=None):
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
    from mindtheedge_amd.models.SemiSupEdgeModel import SemiSupEdgeModel
    from mindtheedge_amd.losses.grad_loss import GradLoss
    K.set_compute_dtype(dtype)
    torch.manual_seed(7)
    net = PackNetSAN01(dropout=dropout, version="1A").cuda()
    model = SemiSupEdgeModel(supervised_loss_weight=1.0, depth_edges_loss_weight=1.0, supervised_method="sparse-silog",
                             supervised_num_scales=1, edges_depth_edge_loss_all_scales=True, flip_lr_prob=0.0)
    model.add_depth_net(net)
    model.add_edge_loss(GradLoss("cross_entropy", True, [], 10.0, 1.0))
    return net, model


def _batch(B, seed=11):
    from mindtheedge_amd.utils.synthetic import synthetic_batch
    return synthetic_batch(B, H, W, seed, torch.device("cuda"))


def _step(dtype, batch, patch=True, side=True):
    from mindtheedge_amd import kernels as K
    K.use_patch_kernels(patch)
    K.use_wgrad_side_stream(side)
    K.set_grad_sink(None)
    try:
        net, model = _model(dtype)
        model.train()
        out = model(batch)
        out["loss"].sum().backward()
        K.join_side_stream()
        torch.cuda.synchronize()
        gsq = {n: float((p.grad.double() ** 2).sum()) for n, p in net.named_parameters() if p.grad is not None}
        return float(out["loss"].detach().sum()), {k: float(v) for k, v in out["metrics"].items()}, gsq
    finally:
        K.use_patch_kernels(True)
        K.use_wgrad_side_stream(True)
        K.set_compute_dtype("bf16")


def test_frames_of_a_batch_are_independent_at_full_size():
    from mindtheedge_amd import kernels as K
    try:
        net, _ = _model("bf16")
        net.eval()
        rgb = _batch(3)["rgb"]
        with torch.no_grad():
            full = [t.clone() for t in net(rgb)["inv_depths"][0]]
            for i in (0, 2):
                one = net(rgb[i:i + 1])["inv_depths"][0]
                for s in range(4):
                    assert tuple(one[s].shape) == (1, 1, H >> s, W >> s)
                    # same bf16 products; only the order of the GroupNorm statistics sums and the split-K / tile
                    # alignment differ between the two launches
                    assert rel_err(one[s].float().cpu(), full[s][i:i + 1].float().cpu()) < 4e-2, (i, s)   # max-norm: a few bf16 ulps after ~60 layers
                    assert float((one[s] - full[s][i:i + 1]).abs().mean()) < 8e-3 * float(full[s][i:i + 1].abs().mean())   # ~2 bf16 ulp
    finally:
        K.set_compute_dtype("bf16")


# Measured on MI355X (profiles/r06_t8_pin.txt): worst max-norm 2.2e-2, worst mean 5.1e-3, worst 32x32-block mean 5.8e-3 of the map's mean value, no map
# bit-identical (the GroupNorm record count and the split-K slab count depend on B: other fp32 summation orders, then ~60 layers of bf16 rounding).
# Bounds: 1.5x the measured max-norm and mean; 2x the measured block mean -- the differences are spread evenly over a map (worst block = 1.15x the map's
# mean), whereas a wrong tile in one corner of one map is a block at O(0.1 .. 1) of the mean value: that is what the block bound catches.
T8_PIN_MAX, T8_PIN_MEAN, T8_PIN_BLOCK = 3.3e-2, 7.7e-3, 1.2e-2


def _frame_vs_batch(one, full, i):
    """(max-norm relative error, mean relative error, bit-identical?, worst 32x32 block) of frame i alone against the same frame inside the batch.
    worst block = the largest mean |difference| over a 32x32-pixel block (16x16 from the third scale on) relative to the map's mean |value|: rounding noise is
    spread over the map, a wrong tile (the failure this test exists for) is not"""
    a, b = one.float(), full[i:i + 1].float()
    blk = 32 if a.shape[-1] >= 640 else 16
    d = torch.nn.functional.avg_pool2d((a - b).abs(), blk)
    return (rel_err(a.cpu(), b.cpu()), float((a - b).abs().mean()) / float(b.abs().mean()), bool(torch.equal(one, full[i:i + 1])),
            float(d.max()) / float(b.abs().mean()))


def test_benchmark_batch_of_8_pins_every_frame_to_the_frame_run_alone():
    """The T8 geometry of BASELINE.json (B = 8, 384x1280, bf16) is the only one in which the >= 200-tile arm of the 8-phase implicit GEMM
    (conv_igemm8_kernel), the tile-count-dependent LDS-patch / GroupNorm routes and the GroupNorm cluster kernels all run; the oracle
    comparisons run at B = 1 / B = 4 (tests/test_gpu_oracle_fullsize.py).  This ties the two together: every frame of a B = 8 eval forward,
    and the four inverse-depth maps of a B = 8 TRAINING-mode forward (dropout off: deterministic), against the same frame run alone.
    Nothing couples the frames (GroupNorm is per sample); what differs between the two launches is the fp32 summation order of the
    GroupNorm records (workgroups per sample depend on B) and of the split-K slabs (splits depend on the tile count), i.e. a few bf16 ulps
    after ~60 layers -- the bound is the one the 3-frame test above holds, and the test reports how many maps came out bit-identical."""
    from mindtheedge_amd import kernels as K
    try:
        net, _ = _model("bf16")
        rgb = _batch(8, seed=23)["rgb"]
        exact = total = 0
        worst = {}
        for mode in ("eval", "train"):
            net.train(mode == "train")
            with torch.no_grad():
                out = net(rgb)["inv_depths"]
                full = [t.clone() for t in (out[0] if mode == "eval" else out)]
                assert len(full) == 4 and all(bool(torch.isfinite(t).all()) for t in full)
                for i in range(8):
                    o = net(rgb[i:i + 1])["inv_depths"]
                    one = o[0] if mode == "eval" else o
                    for s in range(4):
                        assert tuple(one[s].shape) == (1, 1, H >> s, W >> s)
                        mx, mean, same, blk = _frame_vs_batch(one[s], full[s], i)
                        exact += same
                        total += 1
                        w = worst.setdefault(mode, [0.0, 0.0, 0, 0.0])
                        w[0], w[1], w[2], w[3] = max(w[0], mx), max(w[1], mean), w[2] + same, max(w[3], blk)
                        assert mx < T8_PIN_MAX and mean < T8_PIN_MEAN and blk < T8_PIN_BLOCK, (mode, i, s, mx, mean, blk)
        report = "B=8 vs single frame: %d of %d maps bit-identical; " % (exact, total) + "; ".join(
            "%s: worst max-norm %.3e, worst mean %.3e, worst block mean %.3e, %d of 32 bit-identical" % (m, w[0], w[1], w[3], w[2]) for m, w in worst.items())
        print(report)
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(out):                                   # (the measured numbers go on record: profiles/r06_t8_pin.txt)
            with open(os.path.join(out, "t8_pin.txt"), "w") as f:
                f.write(report + "\n")
    finally:
        K.set_compute_dtype("bf16")


def test_full_size_training_step_bf16_vs_fp32_and_kernel_families():
    batch = _batch(1)
    l32, m32, g32 = _step("fp32", batch)
    l16, m16, g16 = _step("bf16", batch)
    assert abs(l16 - l32) <= 1e-3 * abs(l32)                                   # north-star tolerance on the loss scalar
    for k in ("edge_loss", "supervised_loss"):
        assert abs(m16[k] - m32[k]) <= 2e-3 * abs(m32[k]), k
    tot32, tot16 = sum(g32.values()), sum(g16.values())
    assert abs(tot16 - tot32) <= 0.05 * tot32                                  # gradient energy, bf16 storage of ~60 layers
    # LDS-patch conv family vs generic implicit GEMM (bf16, same products)
    lg, _, gg = _step("bf16", batch, patch=False)
    assert abs(lg - l16) <= 2e-4 * abs(l16)
    assert abs(sum(gg.values()) - tot16) <= 0.02 * tot16


def test_full_size_side_stream_schedule_does_not_change_gradients():
    """fp32 mode + gradient sink: the weight-gradient side stream only moves kernels in time."""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.trainers.data_parallel import FlatParameters
    batch = _batch(1)

    def run(side):
        K.use_wgrad_side_stream(side)
        K.set_grad_sink(None)
        try:
            net, model = _model("fp32")
            model.train()
            flat = FlatParameters(net.parameters())
            flat.zero_grad()
            loss = model(batch)["loss"].sum()
            loss.backward()
            K.join_side_stream()
            torch.cuda.synchronize()
            return float(loss.detach()), flat.grad.clone()
        finally:
            K.use_wgrad_side_stream(True)
            K.set_grad_sink(None)
            K.set_compute_dtype("bf16")

    la, ga = run(True)
    lb, gb = run(False)
    assert la == pytest.approx(lb, rel=1e-6)
    assert float((ga - gb).norm()) <= 1e-3 * float(gb.norm())
    assert float(gb.norm()) > 0


def test_high_resolution_768x2560_batch2_training_step_is_finite_and_matches_fp32_loss():
    """BASELINE.json's high-resolution configuration (768x2560, batch 2 per GPU): one bf16 training step at EXACTLY that batch geometry
    (round-3 verdict: only B = 1 had run), every gradient finite, loss within the north-star tolerance of the fp32-mode loss of the same
    kernels, and -- GroupNorm is per sample -- the inverse depth of sample 1 in the batch equal to the same frame run alone."""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.utils.synthetic import synthetic_batch
    batch = synthetic_batch(2, 768, 2560, 5, torch.device("cuda"))
    losses = {}
    try:
        for dtype in ("bf16", "fp32"):
            K.set_grad_sink(None)
            net, model = _model(dtype)
            model.train()
            out = model(batch)
            loss = out["loss"].sum()
            if dtype == "bf16":
                loss.backward()
                K.join_side_stream()
                torch.cuda.synchronize()
                for n, p in net.named_parameters():
                    if p.requires_grad and p.grad is not None:
                        assert bool(torch.isfinite(p.grad).all()), n
                inv = out["inv_depths"][0].detach().float()
                assert tuple(inv.shape) == (2, 1, 768, 2560)
                with torch.no_grad():
                    alone = net(batch["rgb"][1:2])["inv_depths"][0].float()
                # same bf16 products; only the GroupNorm statistics' summation order and the split-K / tile alignment differ between the launches
                assert rel_err(alone[0].cpu(), inv[1].cpu()) < 4e-2
                assert float((alone[0] - inv[1]).abs().mean()) < 8e-3 * float(inv[1].abs().mean())
            losses[dtype] = float(loss.detach())
            del net, model, out, loss
            torch.cuda.empty_cache()
        assert abs(losses["bf16"] - losses["fp32"]) <= 1e-3 * abs(losses["fp32"])
    finally:
        K.set_grad_sink(None)
        K.set_compute_dtype("bf16")
