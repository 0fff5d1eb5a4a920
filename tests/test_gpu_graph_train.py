"""GPU (-m gpu): the training step replayed from HIP graphs (utils/graph.py::GraphedTrainStep) against the eager step:
same losses over several optimisation steps (fp32 mode, no dropout, the flip sequence drawn from the same python RNG
state), Adam's step count / bias corrections advancing through device memory, fresh Dropout2d draws at every replay, a new
batch copied into the static buffers."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dtype, dropout, seed=3):
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
    from mindtheedge_amd.models.SemiSupEdgeModel import SemiSupEdgeModel
    from mindtheedge_amd.losses.grad_loss import GradLoss
    from mindtheedge_amd.trainers.data_parallel import FlatParameters, FusedAdam
    K.set_compute_dtype(dtype)
    K.set_grad_sink(None)
    torch.manual_seed(seed)
    net = PackNetSAN01(dropout=dropout, version="1A").cuda()
    model = SemiSupEdgeModel(supervised_loss_weight=1.0, depth_edges_loss_weight=1.0, supervised_method="sparse-silog",
                             supervised_num_scales=1, edges_depth_edge_loss_all_scales=True, flip_lr_prob=0.5)
    model.add_depth_net(net)
    model.add_edge_loss(GradLoss("cross_entropy", True, [], 10.0, 1.0))
    model.train()
    flat = FlatParameters(net.parameters())
    opt = FusedAdam(flat, lr=1e-3)
    return net, model, opt


def test_graphed_steps_equal_eager_steps():
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.utils.graph import GraphedTrainStep
    from mindtheedge_amd.utils.synthetic import synthetic_batch
    dev = torch.device("cuda")
    batches = [synthetic_batch(2, 64, 128, s, dev) for s in (1, 2, 3, 4, 5)]
    try:
        # eager reference run
        net, model, opt = _setup("fp32", None)
        random.seed(11)
        eager = []
        for b in batches:
            opt.zero_grad()
            out = model(b)
            out["loss"].backward()
            opt.step()
            eager.append(float(out["loss"].detach().sum()))
        p_eager = opt.flatp.flat.clone()
        assert opt.steps == 5
        # graphed run from the same initial weights
        net, model, opt = _setup("fp32", None)
        step = GraphedTrainStep(model, opt, batches[0])
        assert step.graphed, step.error
        assert opt.steps > 0
        # the warm-up steps inside GraphedTrainStep trained the network: start again from the same seed AFTER capture
        torch.manual_seed(3)
        from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
        fresh = PackNetSAN01(dropout=None, version="1A").state_dict()
        with torch.no_grad():
            for n, p in net.named_parameters():
                p.copy_(fresh[n].to(p.device))
        opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.steps = 0
        K.bump_weights_epoch()
        K.prefetch_weight_packs()
        K.join_side_stream()
        random.seed(11)
        got = [float(step(b)["loss"].sum()) for b in batches]
        torch.cuda.synchronize()
        assert opt.steps == 5
        for a, b in zip(got, eager):
            assert a == pytest.approx(b, rel=2e-4), (got, eager)       # Adam at lr 1e-3 amplifies atomics-order noise step by step
        assert got[0] == pytest.approx(eager[0], rel=1e-6)
        assert float((opt.flatp.flat - p_eager).abs().max()) <= 5e-3    # <= a few lr per parameter after 5 steps
        assert len(set(round(x, 4) for x in got)) > 1
    finally:
        K.set_grad_sink(None)
        K.set_compute_dtype("bf16")


def test_replays_draw_fresh_dropout_masks_and_follow_the_flip_draw():
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.utils.graph import GraphedTrainStep
    from mindtheedge_amd.utils.synthetic import synthetic_batch
    batch = synthetic_batch(2, 64, 128, 9, torch.device("cuda"))
    try:
        net, model, opt = _setup("bf16", 0.5)
        for g in opt.param_groups:
            g["lr"] = 0.0                                   # frozen weights: only the dropout masks / the flip change the loss
        state = random.getstate()
        step = GraphedTrainStep(model, opt, batch)
        assert step.graphed, step.error
        assert random.getstate() == state                   # capture does not consume the run's flip draws
        model.flip_lr_prob = 0.0
        a = [float(step(batch)["loss"].sum()) for _ in range(4)]
        assert len(set(a)) == 4, a                          # a new Dropout2d draw per replay
        model.flip_lr_prob = 1.0
        b = float(step(batch)["loss"].sum())
        assert b not in a
    finally:
        K.set_grad_sink(None)
        K.set_compute_dtype("bf16")
