"""CPU: the training-mode statement of the sparse branch in oracle/san_oracle.py (parity unpinned -- MinkowskiEngine is not
available; these tests pin the statement to the semantics it claims: BatchNorm1d over the ACTIVE points, max pooling over active cells)."""
import torch
import torch.nn as nn

from oracle import san_oracle as so


def test_bn_relu_train_is_batchnorm1d_over_the_active_points():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 6, 5, 7, generator=g)
    mask = torch.rand(2, 1, 5, 7, generator=g) < 0.4
    bn = nn.BatchNorm1d(6)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.2, 0.2)
    ref = nn.BatchNorm1d(6)
    ref.load_state_dict(bn.state_dict())
    state = {"weight": bn.weight, "bias": bn.bias, "running_mean": bn.running_mean.clone(), "running_var": bn.running_var.clone(), "eps": bn.eps}
    out = so.bn_relu_train(x, mask, state)
    pts = x.permute(0, 2, 3, 1)[mask[:, 0]]
    want = torch.relu(ref.train()(pts))
    assert torch.allclose(out.permute(0, 2, 3, 1)[mask[:, 0]], want, atol=1e-6)
    assert float(out.detach().permute(0, 2, 3, 1)[~mask[:, 0]].abs().sum()) == 0.0                 # zero off the active set
    assert torch.allclose(state["running_mean"], ref.running_mean, atol=1e-7) and torch.allclose(state["running_var"], ref.running_var, atol=1e-7)


def test_max_pool_gradient_goes_to_the_first_active_maximum():
    feat = torch.zeros(1, 1, 4, 4)
    mask = torch.zeros(1, 1, 4, 4, dtype=torch.bool)
    # coarse cell (1,1): window rows 1..3, columns 1..3; two equal maxima, an inactive larger value must not count
    feat[0, 0, 1, 2] = 0.5; mask[0, 0, 1, 2] = True
    feat[0, 0, 2, 1] = 0.5; mask[0, 0, 2, 1] = True
    feat[0, 0, 3, 3] = 9.0                                                               # inactive
    mask[0, 0, 2, 2] = True; feat[0, 0, 2, 2] = -1.0
    f = feat.clone().requires_grad_(True)
    pooled, m2 = so.max_pool(f, mask)
    assert bool(m2[0, 0, 1, 1]) and float(pooled[0, 0, 1, 1]) == 0.5
    pooled[0, 0, 1, 1].backward()
    assert float(f.grad[0, 0, 1, 2]) == 1.0 and float(f.grad[0, 0, 2, 1]) == 0.0 and float(f.grad[0, 0, 3, 3]) == 0.0


def test_training_statement_is_differentiable_end_to_end():
    from mindtheedge_amd.networks.layers.minkowski_encoder import MinkowskiEncoder
    enc = MinkowskiEncoder([8, 16])
    P = {"mconvs." + k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")))
         for k, v in enc.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    d = (torch.rand(2, 1, 32, 64, generator=g) < 0.15).float() * (1 + torch.rand(2, 1, 32, 64, generator=g))
    feats = so.san_features(P, d, train=True, levels=2)
    assert [tuple(f.shape) for f in feats] == [(2, 8, 16, 32), (2, 16, 8, 16)]
    sum((f * f).sum() for f in feats).backward()
    assert all(p.grad is not None and float(p.grad.abs().sum()) > 0 for k, p in P.items() if p.requires_grad and "layer" in k and "num_batches" not in k)
