"""Golden-vector generator: runs the REAL reference (imported from /root/reference with the
stub set of SURVEY.md 8(c)) on seeded inputs and stores inputs + outputs as small .npz
fixtures next to this file.  Run in the development container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The fixtures are data (inputs and the reference's outputs); no reference source is stored.
Weights come from ``oracle.packnet_oracle.fixture_params`` (deterministic per parameter
name), loaded into the reference modules through ``load_state_dict`` so the fixture does
not depend on the reference's RNG consumption order.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_import  # noqa: E402
from oracle import packnet_oracle as po  # noqa: E402
from oracle import loss_oracle as lo  # noqa: E402


def rnd(name, shape, lo_=-1.0, hi_=1.0):
    u = po.fixture_tensor("input:" + name, shape)           # U[-1,1)
    return (u * 0.5 + 0.5) * (hi_ - lo_) + lo_


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def load_named(module, prefix, P):
    """Load oracle-named params (prefix-stripped) into a reference module."""
    sd = {k[len(prefix) + 1:]: v for k, v in P.items() if k.startswith(prefix + ".")}
    module.load_state_dict(sd, strict=True)


def grads_of(out, G, tensors):
    gs = torch.autograd.grad((out * G).sum(), tensors, allow_unused=True)
    return gs


def layer_fixtures(ns):
    L = ns.layers01
    # ---- N1 Conv2D: cases (name, cin, cout, k, B, H, W)
    for name, cin, cout, k, B, H, W in [("conv2d_k3", 16, 32, 3, 2, 10, 12),
                                        ("conv2d_k5_rgb", 3, 32, 5, 2, 9, 14),
                                        ("conv2d_k7", 32, 32, 7, 1, 12, 16),
                                        ("conv2d_k3_odd65", 65, 32, 3, 2, 8, 10),
                                        ("conv2d_k3_odd193", 193, 128, 3, 1, 6, 8)]:
        spec = po._conv2d_block_spec("m", cin, cout, k)
        P = po.fixture_params(spec, salt=hash_salt(name))
        m = L.Conv2D(cin, cout, k, 1)
        load_named(m, "m", P)
        x = rnd(name + ".x", (B, cin, H, W)).requires_grad_(True)
        y = m(x)
        G = rnd(name + ".G", tuple(y.shape))
        params = list(m.parameters())
        gs = grads_of(y, G, [x] + params)
        save("layer_" + name, x=x, y=y, G=G, dx=gs[0],
             **{"p." + n: P["m." + n] for n, _ in m.named_parameters()},
             **{"g." + n: g for (n, _), g in zip(m.named_parameters(), gs[1:])})
    # ---- N2 ResidualConv (no dropout) + with an explicit channel mask emulated outside
    for name, cin, cout, B, H, W in [("resconv_32_64", 32, 64, 2, 8, 10), ("resconv_64_64", 64, 64, 1, 6, 12)]:
        spec = (po._conv2d_block_spec("m.conv1", cin, cout, 3) + po._conv2d_block_spec("m.conv2", cout, cout, 3) +
                [("m.conv3.weight", (cout, cin, 1, 1)), ("m.conv3.bias", (cout,)),
                 ("m.normalize.weight", (cout,)), ("m.normalize.bias", (cout,))])
        P = po.fixture_params(spec, salt=hash_salt(name))
        m = L.ResidualConv(cin, cout, 1, dropout=None)
        load_named(m, "m", P)
        x = rnd(name + ".x", (B, cin, H, W)).requires_grad_(True)
        y = m(x)
        G = rnd(name + ".G", tuple(y.shape))
        gs = grads_of(y, G, [x] + list(m.parameters()))
        save("layer_" + name, x=x, y=y, G=G, dx=gs[0],
             **{"p." + n: P["m." + n] for n, _ in m.named_parameters()},
             **{"g." + n: g for (n, _), g in zip(m.named_parameters(), gs[1:])})
    # ---- N5 InvDepth
    for name, cin, B, H, W in [("invdepth_32", 32, 2, 8, 12), ("invdepth_256", 256, 1, 4, 6)]:
        spec = [("m.conv1.weight", (1, cin, 3, 3)), ("m.conv1.bias", (1,))]
        P = po.fixture_params(spec, salt=hash_salt(name))
        m = L.InvDepth(cin)
        load_named(m, "m", P)
        x = rnd(name + ".x", (B, cin, H, W)).requires_grad_(True)
        y = m(x)
        G = rnd(name + ".G", tuple(y.shape))
        gs = grads_of(y, G, [x] + list(m.parameters()))
        save("layer_" + name, x=x, y=y, G=G, dx=gs[0],
             **{"p." + n: P["m." + n] for n, _ in m.named_parameters()},
             **{"g." + n: g for (n, _), g in zip(m.named_parameters(), gs[1:])})
    # ---- packing / pixel shuffle index maps
    x = rnd("packing.x", (2, 3, 6, 8))
    save("layer_packing", x=x, y=L.packing(x), z=torch.nn.PixelShuffle(2)(L.packing(x)))
    # ---- N3 PackLayerConv3d, N4 UnpackLayerConv3d
    for name, c, k, B, H, W in [("pack3d_c16_k5", 16, 5, 2, 8, 12), ("pack3d_c32_k3", 32, 3, 1, 8, 8)]:
        spec = (po._conv2d_block_spec("m.conv", c * 16, c, k) +
                [("m.conv3d.weight", (4, 1, 3, 3, 3)), ("m.conv3d.bias", (4,))])
        P = po.fixture_params(spec, salt=hash_salt(name), bias_scale=0.2)
        m = L.PackLayerConv3d(c, k, d=4)
        load_named(m, "m", P)
        x = rnd(name + ".x", (B, c, H, W)).requires_grad_(True)
        y = m(x)
        G = rnd(name + ".G", tuple(y.shape))
        gs = grads_of(y, G, [x] + list(m.parameters()))
        save("layer_" + name, x=x, y=y, G=G, dx=gs[0],
             **{"p." + n: P["m." + n] for n, _ in m.named_parameters()},
             **{"g." + n: g for (n, _), g in zip(m.named_parameters(), gs[1:])})
    for name, cin, cout, B, H, W in [("unpack3d_64_32", 64, 32, 2, 6, 8), ("unpack3d_32_16", 32, 16, 1, 5, 7)]:
        spec = (po._conv2d_block_spec("m.conv", cin, cout, 3) +
                [("m.conv3d.weight", (4, 1, 3, 3, 3)), ("m.conv3d.bias", (4,))])
        P = po.fixture_params(spec, salt=hash_salt(name), bias_scale=0.2)
        m = L.UnpackLayerConv3d(cin, cout, 3, d=4)
        load_named(m, "m", P)
        x = rnd(name + ".x", (B, cin, H, W)).requires_grad_(True)
        y = m(x)
        G = rnd(name + ".G", tuple(y.shape))
        gs = grads_of(y, G, [x] + list(m.parameters()))
        save("layer_" + name, x=x, y=y, G=G, dx=gs[0],
             **{"p." + n: P["m." + n] for n, _ in m.named_parameters()},
             **{"g." + n: g for (n, _), g in zip(m.named_parameters(), gs[1:])})


def hash_salt(name):
    return sum(ord(c) * (i + 1) for i, c in enumerate(name)) % 100003


def loss_fixtures(ns):
    # ---- L1 inv2depth / depth2inv incl. clamp and zero handling
    inv = torch.tensor([0.0, 1e-7, 1e-6, 0.5, 2.0, -1.0]).view(1, 1, 2, 3)
    dep = torch.tensor([0.0, -2.0, 1e-7, 1.0, 80.0, 3.5]).view(1, 1, 2, 3)
    save("loss_inv_depth", inv=inv, depth_of_inv=ns.inv2depth(inv), dep=dep, inv_of_dep=ns.depth2inv(dep))
    # ---- L2 GradLayer: no normals, random normals, normals exactly on bin edges
    gl = ns.GradLayer()
    x = rnd("gradlayer.x", (2, 1, 12, 16), 0.5, 40.0)
    mag, xv, xh = gl(x, None)
    nrm = rnd("gradlayer.n", (2, 1, 12, 16), -math.pi, math.pi)
    mag_n, _, _ = gl(x, nrm)
    edges = torch.tensor([k * np.pi / 8 for k in range(-8, 9)], dtype=torch.float32)     # float32(k*pi/8)
    edges = torch.cat([edges, torch.nextafter(edges, torch.tensor(10.0)), torch.nextafter(edges, torch.tensor(-10.0)),
                       torch.tensor([-math.pi, math.pi, 0.0, 3.2, -3.2])])
    ne = edges[torch.arange(2 * 12 * 16) % edges.numel()].view(2, 1, 12, 16).contiguous()
    mag_e, _, _ = gl(x, ne)
    save("loss_gradlayer", x=x, mag=mag, xv=xv, xh=xh, normal=nrm, mag_n=mag_n, normal_edges=ne, mag_e=mag_e)
    # ---- L3/L4 GradLoss (cross_entropy, weight 10, pos_to_neg 1): soft labels; mask None / binary / all-ones /
    #      all-negative batch; fwd loss + d loss / d depth
    head = ns.GradLoss("cross_entropy", True, [], 10.0, 1.0)
    B, H, W = 3, 16, 24
    inv = rnd("gradloss.inv", (B, 1, H, W), 0.02, 1.9)
    on = (rnd("gradloss.on", (B, 1, H, W), 0, 1) < 0.08).float()
    edge = on * rnd("gradloss.e", (B, 1, H, W), 0.0, 1.0)
    nrm = rnd("gradloss.n", (B, 1, H, W), -math.pi, math.pi)
    mask = (rnd("gradloss.m", (B, 1, H, W), 0, 1) < 0.7).float()
    out = {}
    cases = {"nomask": (edge, None, nrm), "binmask": (edge, mask, nrm), "onesmask": (edge, torch.ones_like(mask), nrm),
             "allneg": (torch.zeros_like(edge), None, nrm), "nonormal": (edge, None, None),
             "allpos": (torch.ones_like(edge), None, nrm)}
    for cname, (e, m, n) in cases.items():
        d = ns.inv2depth(inv).detach().requires_grad_(True)
        loss, g = head(d, e, m, True, True, 4, n)
        (dd,) = torch.autograd.grad(loss, d)
        out["loss_" + cname], out["g_" + cname], out["ddepth_" + cname] = loss, g, dd
    # probability-input variant (is_grad=False, is_sigmoid=False) used by the DEE model
    pr = rnd("gradloss.p", (B, 1, H, W), 0.0, 1.0).requires_grad_(True)
    lp, _ = head(pr, edge, None, False, False, 4, None)
    out["loss_prob"], out["dprob"] = lp, torch.autograd.grad(lp, pr)[0]
    # bilinear-resize branch (prediction at half resolution)
    dh = ns.inv2depth(rnd("gradloss.invh", (B, 1, H // 2, W // 2), 0.02, 1.9)).requires_grad_(True)
    lh, _ = head(dh, edge, None, True, True, 4, nrm)
    out["depth_half"], out["loss_half"], out["ddepth_half"] = dh, lh, torch.autograd.grad(lh, dh)[0]
    save("loss_gradloss", inv=inv, edge=edge, normal=nrm, mask=mask, prob=pr, **out)
    # ---- L6 silog / supervised loss (+ empty-mask case)
    sup = ns.SupervisedLoss(supervised_method="sparse-silog", supervised_num_scales=1)
    inv0 = rnd("silog.inv", (2, 1, 16, 24), 0.02, 1.9).requires_grad_(True)
    dens = (rnd("silog.on", (2, 1, 16, 24), 0, 1) < 0.1).float()
    depth = dens * rnd("silog.d", (2, 1, 16, 24), 1.0, 80.0)
    o = sup([inv0, inv0[:, :, ::2, ::2]], ns.depth2inv(depth))
    (dinv,) = torch.autograd.grad(o["loss"].sum(), inv0)
    inv0b = inv0.detach().clone()
    o_empty = sup([inv0b], ns.depth2inv(torch.zeros_like(depth)))
    save("loss_silog", inv=inv0, depth=depth, loss=o["loss"], dinv=dinv, loss_empty=o_empty["loss"])


def net_fixtures(ns):
    P = po.fixture_params()
    B, H, W = 2, 64, 128
    for drop_tag, dropout in (("", None),):
        net = ns.PackNetSAN01(dropout=dropout, version="1A")
        net.is_depth_aux_net = False                       # SURVEY headline fact: never assigned upstream
        missing = net.load_state_dict(P, strict=True)
        rgb = rnd("net.rgb", (B, 3, H, W), 0.0, 1.0)
        net.train()
        out = net(rgb)["inv_depths"]
        net.eval()
        with torch.no_grad():
            oe = net(rgb)["inv_depths"]
        save("net_packnetsan01_64x128", rgb=rgb,
             **{"train_inv%d" % i: t for i, t in enumerate(out)},
             **{"eval_inv%d" % i: t for i, t in enumerate(oe[0])},
             **{"eval_feat%d_mean_abs" % i: t.abs().mean() for i, t in enumerate(oe[1])},
             **{"eval_feat%d_corner" % i: t[:, :4, :3, :3] for i, t in enumerate(oe[1])})
        # ---- L7 model-level: SemiSupEdgeModel loss + parameter gradients
        model = ns.SemiSupEdgeModel(supervised_loss_weight=1.0, depth_edges_loss_weight=1.0,
                                    supervised_method="sparse-silog", supervised_num_scales=1,
                                    edges_depth_edge_loss_all_scales=True, upsample_depth_maps=False,
                                    flip_lr_prob=0.0)
        model.add_depth_net(net)
        model.add_edge_loss(ns.GradLoss("cross_entropy", True, [], 10.0, 1.0))
        model.train()
        batch = lo.synthetic_batch(B, H, W, seed=7)
        batch["edge"] = batch["edge"] * 1.0
        net.zero_grad()
        o = model(dict(batch))
        o["loss"].backward()
        grads = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
        keep = ["encoder.pre_calc.conv_base.weight", "encoder.pre_calc.normalize.weight",
                "encoder.pack1.conv3d.weight", "encoder.pack1.conv3d.bias", "encoder.conv2.0.conv3.bias",
                "encoder.pack5.conv.normalize.bias", "decoder.unpack1.conv3d.weight", "decoder.iconv1.conv_base.bias",
                "decoder.disp1_layer.conv1.weight", "decoder.disp4_layer.conv1.weight", "decoder.iconv3.normalize.weight"]
        names = sorted(grads)
        save("model_semisup_64x128",
             **{"batch." + k: v for k, v in batch.items()},
             loss=o["loss"].detach(), edge_loss=o["metrics"]["edge_loss"], supervised_loss=o["metrics"]["supervised_loss"],
             grad_names=np.array(names), grad_sumsq=np.array([float((grads[n].double() ** 2).sum()) for n in names]),
             grad_sum=np.array([float(grads[n].double().sum()) for n in names]),
             **{"grad." + n: grads[n] for n in keep})
        # flipped run (H1): force the flip branch deterministically
        model.flip_lr_prob = 1.0
        of = model(dict(batch))
        save("model_semisup_64x128_flip", loss=of["loss"].detach(), edge_loss=of["metrics"]["edge_loss"],
             supervised_loss=of["metrics"]["supervised_loss"])
        # Adam step reference on three tensors (H2)
        ps = [torch.nn.Parameter(rnd("adam.p%d" % i, s)) for i, s in enumerate([(7,), (3, 5), (2, 3, 3, 3)])]
        opt = torch.optim.Adam(ps, lr=1e-4)
        traj = {}
        for step in range(3):
            for i, p in enumerate(ps):
                p.grad = rnd("adam.g%d.%d" % (i, step), tuple(p.shape)) * (10.0 ** (i - 1))
                traj["g%d_s%d" % (i, step)] = p.grad.clone()
            opt.step()
            for i, p in enumerate(ps):
                traj["p%d_s%d" % (i, step)] = p.detach().clone()
        save("adam_steps", **{"p%d_init" % i: rnd("adam.p%d" % i, tuple(p.shape)) for i, p in enumerate(ps)}, **traj)


if __name__ == "__main__":
    assert ref_import.reference_available(), "run in the development container (needs /root/reference)"
    torch.set_num_threads(8)
    torch.manual_seed(0)
    ns = ref_import.import_reference()
    layer_fixtures(ns)
    loss_fixtures(ns)
    net_fixtures(ns)
