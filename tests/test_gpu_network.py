"""GPU parity (-m gpu): whole PackNetSAN01 network and SemiSupEdgeModel training loss / gradients against the golden
vectors of the reference at 64x128 (B=2)."""
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _net(dtype):
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
    from oracle import packnet_oracle as po
    K.set_compute_dtype(dtype)
    net = PackNetSAN01(dropout=None, version="1A")
    net.load_state_dict(po.fixture_params(), strict=True)
    return net.cuda()


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 5e-2)])
def test_network_outputs(dtype, tol):
    from mindtheedge_amd import kernels as K
    g = load_golden("net_packnetsan01_64x128")
    net = _net(dtype)
    try:
        net.train()
        out = net(g["rgb"].cuda())["inv_depths"]
        for i in range(4):
            assert tuple(out[i].shape) == tuple(g["train_inv%d" % i].shape)
            assert rel_err(out[i].cpu(), g["train_inv%d" % i]) < tol, i
        net.eval()
        with torch.no_grad():
            oe = net(g["rgb"].cuda())["inv_depths"]
        assert len(oe) == 2 and len(oe[0]) == 4 and len(oe[1]) == 6
        for i, f in enumerate(oe[1]):
            assert rel_err(f.float().cpu()[:, :4, :3, :3], g["eval_feat%d_corner" % i]) < 20 * tol
    finally:
        K.set_compute_dtype("bf16")


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 5e-2)])
def test_model_loss_and_gradients(dtype, tol):
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.models.SemiSupEdgeModel import SemiSupEdgeModel
    from mindtheedge_amd.losses.grad_loss import GradLoss
    g = load_golden("model_semisup_64x128")
    net = _net(dtype)
    try:
        model = SemiSupEdgeModel(supervised_loss_weight=1.0, depth_edges_loss_weight=1.0, supervised_method="sparse-silog",
                                 supervised_num_scales=1, edges_depth_edge_loss_all_scales=True, upsample_depth_maps=False,
                                 flip_lr_prob=0.0)
        model.add_depth_net(net)
        model.add_edge_loss(GradLoss("cross_entropy", True, [], 10.0, 1.0))
        model.train()
        batch = {k[6:]: v.cuda() for k, v in g.items() if k.startswith("batch.")}
        out = model(batch)
        assert rel_err(out["loss"].cpu(), g["loss"]) < tol
        assert rel_err(out["metrics"]["edge_loss"].cpu(), g["edge_loss"]) < tol
        assert rel_err(out["metrics"]["supervised_loss"].cpu(), g["supervised_loss"]) < tol
        out["loss"].sum().backward()
        grads = dict(net.named_parameters())
        gtol = 5e-3 if dtype == "fp32" else 0.25
        for k, v in g.items():
            if k.startswith("grad."):
                assert rel_err(grads[k[5:]].grad.cpu(), v) < gtol, k
        names = [str(n) for n in g["grad_names"]]
        bad = []
        for n, ss in zip(names, g["grad_sumsq"].tolist()):
            got = float((grads[n].grad.double() ** 2).sum())
            if abs(got - ss) > 2 * gtol * max(ss, 1e-10):
                bad.append((n, got, ss))
        assert not bad, bad[:8]
        # H1: forced whole-batch flip gives the reference's flipped-run loss
        gf = load_golden("model_semisup_64x128_flip")
        model.flip_lr_prob = 1.0
        of = model(batch)
        assert rel_err(of["loss"].detach().cpu(), gf["loss"]) < tol
    finally:
        K.set_compute_dtype("bf16")
