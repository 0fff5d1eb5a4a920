"""Pins the CPU oracle (oracle/) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py, run in the development container).  CPU only."""
import math

import pytest
import torch

from conftest import load_golden, rel_err
from oracle import loss_oracle as lo
from oracle import packnet_oracle as po

TOL = 2e-5   # fp32 reference vs fp32 oracle, same op order up to reassociation


def _params(g, prefix="m."):
    return {prefix + k[2:]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith("p.")}


def _check_layer(g, fn):
    P = _params(g)
    x = g["x"].clone().requires_grad_(True)
    y = fn(x, P)
    assert rel_err(y, g["y"]) < TOL
    names = sorted(P)
    grads = torch.autograd.grad((y * g["G"]).sum(), [x] + [P[n] for n in names])
    assert rel_err(grads[0], g["dx"]) < 5 * TOL
    for n, gr in zip(names, grads[1:]):
        ref = g["g." + n[2:]]
        # a conv bias in front of a 1-channel-per-group GroupNorm has an exactly-zero gradient: allow abs noise
        assert rel_err(gr, ref) < 5 * TOL or float((gr - ref).abs().max()) < 1e-5, n


@pytest.mark.parametrize("name", ["conv2d_k3", "conv2d_k5_rgb", "conv2d_k7", "conv2d_k3_odd65", "conv2d_k3_odd193"])
def test_conv_gn_elu(name):
    _check_layer(load_golden("layer_" + name), lambda x, P: po.conv_gn_elu(x, P, "m"))


@pytest.mark.parametrize("name", ["resconv_32_64", "resconv_64_64"])
def test_residual_conv(name):
    _check_layer(load_golden("layer_" + name), lambda x, P: po.residual_conv(x, P, "m"))


@pytest.mark.parametrize("name", ["invdepth_32", "invdepth_256"])
def test_inv_depth_head(name):
    _check_layer(load_golden("layer_" + name), lambda x, P: po.inv_depth_head(x, P, "m"))


def test_packing_is_inverse_of_pixel_shuffle():
    g = load_golden("layer_packing")
    assert torch.equal(po.packing(g["x"]), g["y"])
    assert torch.equal(torch.nn.functional.pixel_shuffle(po.packing(g["x"]), 2), g["x"])
    assert torch.equal(g["z"], g["x"])


@pytest.mark.parametrize("name", ["pack3d_c16_k5", "pack3d_c32_k3"])
def test_pack_conv3d(name):
    _check_layer(load_golden("layer_" + name), lambda x, P: po.pack_conv3d(x, P, "m"))


@pytest.mark.parametrize("name", ["unpack3d_64_32", "unpack3d_32_16"])
def test_unpack_conv3d(name):
    _check_layer(load_golden("layer_" + name), lambda x, P: po.unpack_conv3d(x, P, "m"))


def test_inv_depth_conversions():
    g = load_golden("loss_inv_depth")
    assert torch.equal(lo.inv2depth(g["inv"]), g["depth_of_inv"])
    assert torch.equal(lo.depth2inv(g["dep"]), g["inv_of_dep"])


def test_grad_layer_all_variants():
    g = load_golden("loss_gradlayer")
    assert rel_err(lo.grad_layer(g["x"], None), g["mag"]) < 1e-6
    assert rel_err(lo.grad_layer(g["x"], g["normal"]), g["mag_n"]) < 1e-6
    # angles exactly on / one ulp around the k*pi/8 bin edges must select the same kernel
    assert torch.equal(lo.grad_layer(g["x"], g["normal_edges"]), g["mag_e"])


@pytest.mark.parametrize("case", ["nomask", "binmask", "onesmask", "allneg", "nonormal", "allpos"])
def test_grad_loss_cases(case):
    g = load_golden("loss_gradloss")
    edge = {"allneg": torch.zeros_like(g["edge"]), "allpos": torch.ones_like(g["edge"])}.get(case, g["edge"])
    mask = {"binmask": g["mask"], "onesmask": torch.ones_like(g["mask"])}.get(case)
    normal = None if case == "nonormal" else g["normal"]
    d = lo.inv2depth(g["inv"]).requires_grad_(True)
    loss, gmap = lo.grad_loss(d, edge, mask, True, True, 4, normal)
    assert rel_err(loss, g["loss_" + case]) < TOL
    assert rel_err(gmap, g["g_" + case]) < 1e-6
    (dd,) = torch.autograd.grad(loss, d)
    assert rel_err(dd, g["ddepth_" + case]) < 5 * TOL


def test_grad_loss_probability_input_and_resize():
    g = load_golden("loss_gradloss")
    p = g["prob"].clone().requires_grad_(True)
    loss, _ = lo.grad_loss(p, g["edge"], None, False, False, 4, None)
    assert rel_err(loss, g["loss_prob"]) < TOL
    assert rel_err(torch.autograd.grad(loss, p)[0], g["dprob"]) < 5 * TOL
    dh = g["depth_half"].clone().requires_grad_(True)
    lh, _ = lo.grad_loss(dh, g["edge"], None, True, True, 4, g["normal"])
    assert rel_err(lh, g["loss_half"]) < TOL
    assert rel_err(torch.autograd.grad(lh, dh)[0], g["ddepth_half"]) < 5 * TOL


def test_silog_supervised_loss():
    g = load_golden("loss_silog")
    inv = g["inv"].clone().requires_grad_(True)
    loss = lo.supervised_silog_loss(inv, g["depth"])
    assert rel_err(loss, g["loss"]) < TOL
    assert rel_err(torch.autograd.grad(loss, inv)[0], g["dinv"]) < 5 * TOL
    empty = lo.supervised_silog_loss(g["inv"], torch.zeros_like(g["depth"]))
    assert math.isnan(float(empty)) and math.isnan(float(g["loss_empty"]))     # mean of empty -> nan, both sides


def test_adam_reference_steps():
    g = load_golden("adam_steps")
    for i in range(3):
        p = g["p%d_init" % i].clone()
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for step in range(3):
            p, m, v = lo.adam_step(p, g["g%d_s%d" % (i, step)], m, v, step + 1)
            assert rel_err(p, g["p%d_s%d" % (i, step)]) < 1e-6


def test_param_spec_matches_reference_inventory():
    spec = po.param_spec()
    assert sum(int(torch.tensor(s).prod()) for _, s in spec) == 76997806
    assert len(spec) == 218                                                   # 218 parameter tensors
    assert ("encoder.conv2.0.conv3.0.weight", (64, 32, 1, 1)) in po.param_spec(dropout=0.5)


def test_network_and_model_fixtures():
    gn = load_golden("net_packnetsan01_64x128")
    P = {k: v.requires_grad_(True) for k, v in po.fixture_params().items()}
    out = po.packnet_san01(gn["rgb"], P, training=True)["inv_depths"]
    for i in range(4):
        assert rel_err(out[i], gn["train_inv%d" % i]) < 1e-4
        assert rel_err(out[i], gn["eval_inv%d" % i]) < 1e-4          # dropout=None: eval == train
    with torch.no_grad():
        feats = po.packnet_san01(gn["rgb"], P, training=False)["inv_depths"][1]
    for i, f in enumerate(feats):
        assert rel_err(f[:, :4, :3, :3], gn["eval_feat%d_corner" % i]) < 1e-3
        assert abs(float(f.abs().mean()) - float(gn["eval_feat%d_mean_abs" % i])) < 1e-4
    gm = load_golden("model_semisup_64x128")
    batch = {k[6:]: v for k, v in gm.items() if k.startswith("batch.")}
    assert torch.equal(batch["rgb"], lo.synthetic_batch(2, 64, 128, seed=7)["rgb"])
    out = po.packnet_san01(batch["rgb"], P, training=True)["inv_depths"]
    res = lo.semisup_edge_model_loss(out, batch)
    assert rel_err(res["loss"], gm["loss"]) < 1e-4
    assert rel_err(res["edge_loss"], gm["edge_loss"]) < 1e-4
    assert rel_err(res["supervised_loss"], gm["supervised_loss"]) < 1e-4
    res["loss"].sum().backward()
    names = [str(n) for n in gm["grad_names"]]
    for n, ss in zip(names, gm["grad_sumsq"].tolist()):
        got = float((P[n].grad.double() ** 2).sum())
        assert abs(got - ss) <= 2e-3 * max(ss, 1e-12), n
    for k, v in gm.items():
        if k.startswith("grad."):
            assert rel_err(P[k[5:]].grad, v) < 2e-3, k
    # H1: whole-batch flip of rgb in, outputs flipped back (model_utils.py:98-151)
    gf = load_golden("model_semisup_64x128_flip")
    with torch.no_grad():
        outf = [lo.flip_lr(t) for t in po.packnet_san01(lo.flip_lr(batch["rgb"]), P, training=True)["inv_depths"]]
        resf = lo.semisup_edge_model_loss(outf, batch)
    assert rel_err(resf["loss"], gf["loss"]) < 1e-4
