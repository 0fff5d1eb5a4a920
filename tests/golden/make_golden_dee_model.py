"""Generates tests/golden/model_dee_rgb_64x128.npz from the upstream reference (development container only): one RGB-only
training step of EdgeEstimationLIDARModel (packnet_code/packnet_sfm/models/EdgeEstimationLIDARModel.py:87-181, batch
without 'input_depth') on the fixture weights -- loss, metric and parameter-gradient checksums.

    python tests/golden/make_golden_dee_model.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_import                                   # noqa: E402
from oracle import packnet_oracle as po             # noqa: E402
from oracle import loss_oracle as lo                # noqa: E402


def main():
    assert ref_import.reference_available()
    ns = ref_import.import_reference()
    from packnet_code.packnet_sfm.models.EdgeEstimationLIDARModel import EdgeEstimationLIDARModel
    torch.set_num_threads(8)
    P = po.fixture_params()
    B, H, W = 2, 64, 128
    net = ns.PackNetSAN01(dropout=None, version="1A")
    net.is_depth_aux_net = False
    net.load_state_dict(P, strict=True)
    model = EdgeEstimationLIDARModel(supervised_loss_weight=0.0, weight_rgbd=1.0, edges_depth_edge_loss_all_scales=True,
                                     upsample_depth_maps=False, flip_lr_prob=0.0)
    model.add_depth_net(net)
    model.add_edge_loss(ns.GradLoss("cross_entropy", True, [], 10.0, 1.0))
    model.train()
    batch = lo.synthetic_batch(B, H, W, seed=13)
    batch = {k: v for k, v in batch.items() if not k.startswith("normal") and k != "depth"}
    net.zero_grad()
    o = model(dict(batch))
    o["loss"].sum().backward()
    grads = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    names = sorted(grads)
    keep = ["encoder.pre_calc.conv_base.weight", "decoder.disp1_layer.conv1.weight", "decoder.disp4_layer.conv1.bias", "decoder.iconv1.normalize.weight"]
    out = {"batch." + k: v for k, v in batch.items()}
    out.update(loss=o["loss"].detach(), edge_loss=o["metrics"]["edge_loss"], prob0=o["inv_depths"][0].detach(), prob3=o["inv_depths"][3].detach(),
               grad_names=np.array(names), grad_sumsq=np.array([float((grads[n].double() ** 2).sum()) for n in names]),
               grad_sum=np.array([float(grads[n].double().sum()) for n in names]))
    out.update({"grad." + n: grads[n] for n in keep})
    np.savez_compressed(os.path.join(HERE, "model_dee_rgb_64x128.npz"),
                        **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print("loss", o["loss"].detach(), "edge_loss", o["metrics"]["edge_loss"], "params with grad", len(names))


if __name__ == "__main__":
    main()
