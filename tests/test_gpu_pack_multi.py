"""GPU (-m gpu): the multi-tensor weight pack (mte_pack_conv_weights_multi: every forward / data-gradient / LDS-patch fragment pack
of the network in three launches after an optimizer step) writes exactly the bytes of the per-layer entry points it replaces
(mte_pack_conv_weights + mte_conv2d_patch_repack) -- reference: the weights of nn.Conv2d in Conv2D / ResidualConv / InvDepth,
networks/layers/packnet/layers01.py:29,61,116, re-read by every forward pass after optimizer.step()."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(32, 3, 5), (32, 32, 7), (64, 1024, 3), (32, 512, 5), (64, 64, 1), (128, 200, 3), (512, 768, 3), (72, 32, 3), (16, 16, 3),
          (32, 65, 3), (256, 4096, 3)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_multi_tensor_pack_equals_per_layer_packs(dtype):
    from mindtheedge_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    refs, news, params = [], [], []
    for co, ci, k in SHAPES:
        w = torch.randn(co, ci, k, k, generator=g).cuda()
        ref = K.WeightPack()
        wf, wb = ref.get(w, dtype, True)
        patch = dtype == torch.bfloat16 and co <= 64 and K.round8(ci) <= 1024
        want = [wf.clone(), wb.clone(), ref.get_patch(w, 'f').clone() if patch else None, ref.get_patch(w, 'b').clone() if patch else None]
        # the pack under test is built on OTHER values through the per-layer path, then refreshed by the multi-tensor kernel
        w2 = torch.randn(co, ci, k, k, generator=g).cuda()
        new = K.WeightPack()
        new.get(w2, dtype, True)
        if patch:
            new.get_patch(w2, 'f')
            new.get_patch(w2, 'b')
        w2.copy_(w)
        refs.append(want)
        news.append((new, patch))
        params.append((w, w2, ref))
    K.bump_weights_epoch()
    K.prefetch_weight_packs()
    torch.cuda.synchronize()
    for (new, patch), want, (w, w2, ref), shape in zip(news, refs, params, SHAPES):
        assert new._event is not None                              # refreshed by the prefetch, not lazily
        wf2, wb2 = new.get(w2, dtype, True)
        torch.cuda.synchronize()
        assert torch.equal(wf2, want[0]), shape
        assert torch.equal(wb2, want[1]), shape
        if patch:
            assert new.pf_ok and new.pb_ok
            assert torch.equal(new.get_patch(w2, 'f'), want[2]), shape
            assert torch.equal(new.get_patch(w2, 'b'), want[3]), shape


def test_network_step_uses_the_multi_tensor_pack():
    """a training step of the real network: after FusedAdam.step() every pack the next forward needs carries a prefetch event and
    the next step's loss equals the loss of a run with the side stream off (the same launches on the main stream, no events)"""
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
    from mindtheedge_amd.models.SemiSupEdgeModel import SemiSupEdgeModel
    from mindtheedge_amd.losses.grad_loss import GradLoss
    from mindtheedge_amd.trainers.data_parallel import FlatParameters, FusedAdam
    from mindtheedge_amd.utils.synthetic import synthetic_batch

    def run(side):
        K.set_grad_sink(None)
        K.use_wgrad_side_stream(side)
        torch.manual_seed(7)
        net = PackNetSAN01(dropout=None, version="1A").cuda()
        model = SemiSupEdgeModel(supervised_loss_weight=1.0, depth_edges_loss_weight=1.0, supervised_method="sparse-silog",
                                 supervised_num_scales=1, edges_depth_edge_loss_all_scales=True, flip_lr_prob=0.0)
        model.add_depth_net(net)
        model.add_edge_loss(GradLoss("cross_entropy", True, [], 10.0, 1.0))
        model.train()
        flat = FlatParameters(net.parameters())
        opt = FusedAdam(flat, lr=1e-3)
        batch = synthetic_batch(2, 64, 128, 3, torch.device("cuda"))
        losses = []
        for _ in range(3):
            opt.zero_grad()
            out = model(batch)
            out["loss"].backward()
            opt.step()
            losses.append(float(out["loss"].sum()))
        packs = [m.pack for m in net.modules() if isinstance(getattr(m, "pack", None), K.WeightPack)]     # (this network's packs only)
        n_ev = sum(1 for pk in packs if pk._event is not None)
        torch.cuda.synchronize()
        return losses, n_ev

    try:
        K.set_compute_dtype("fp32")
        a, n_a = run(True)
        b, n_b = run(False)
        assert n_a >= 40 and n_b == 0                            # every conv layer of the network (47 + the stem)
        assert a[0] == pytest.approx(b[0], rel=1e-6)
        for x, y in zip(a, b):
            assert x == pytest.approx(y, rel=2e-4)               # (backward atomics order; the packs themselves are bit-equal)
        assert len(set(a)) == 3                                      # the weights did move
    finally:
        K.use_wgrad_side_stream(True)
        K.set_grad_sink(None)
        K.set_compute_dtype("bf16")
