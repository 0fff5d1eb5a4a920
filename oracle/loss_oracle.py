"""CPU ORACLE (test infrastructure, NOT product code) -- depth-edge loss, silog loss, model loss.

Plain PyTorch-CPU restatement (autograd-differentiable, dtype-generic so tests can run it
in float64) of the loss side of the hot path.  Pinned against the imported reference by
``tests/golden/*.npz`` (see ``tests/golden/make_golden.py``).  Never imported by
``mindtheedge_amd``.

Reference (relative to /root/reference/packnet_code/packnet_sfm):
  utils/depth.py:104-144             inv2depth / depth2inv
  losses/grad_loss.py:20-31,65-95    GradLayer (4 Sobel kernels, normal-selected direction)
  losses/grad_loss.py:122-159        GradLoss.forward
  losses/grad_loss.py:161-219        GradLoss.comp_cross_entropy
  losses/supervised_loss.py:57-69    SilogLoss
  losses/supervised_loss.py:155-216  SupervisedLoss.calculate_loss / forward
  models/SemiSupEdgeModel.py:98-198  loss composition
  models/model_utils.py:98-151, models/SfmModel.py:58-96   whole-batch horizontal flip
"""
import math

import torch
import torch.nn.functional as F

SOBEL = {
    "v": [[-1, -2, -1], [0, 0, 0], [1, 2, 1]],
    "h": [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]],
    "lr": [[-2, -1, 0], [-1, 0, 1], [0, 1, 2]],
    "rl": [[0, 1, 2], [-1, 0, 1], [-2, -1, 0]],
}
PI_8 = math.pi / 8


def inv2depth(inv):
    return 1.0 / inv.clamp(min=1e-6)


def depth2inv(depth):
    inv = 1.0 / depth.clamp(min=1e-6)
    return torch.where(depth <= 0.0, torch.zeros_like(inv), inv)


def _corr3(x, name):
    k = torch.tensor(SOBEL[name], dtype=x.dtype).view(1, 1, 3, 3)
    return F.conv2d(x, k, padding=1)


def direction_code(normal):
    """0=h (default) 1=v 2=rl 3=lr, half-open bins in float32-compare semantics of
    grad_loss.py:80-93 (thresholds are python doubles k*pi/8 compared against the tensor)."""
    n = normal
    code = torch.zeros_like(n, dtype=torch.int64)
    in_ = lambda lo, hi: (n >= lo * PI_8) & (n < hi * PI_8)
    code = torch.where(in_(-5, -3) | in_(3, 5), torch.full_like(code, 1), code)
    code = torch.where(in_(-7, -5) | in_(1, 3), torch.full_like(code, 2), code)
    code = torch.where(in_(-3, -1) | in_(5, 7), torch.full_like(code, 3), code)
    return code


def grad_layer(x, normal=None):
    """Returns the edge-strength map g (grad_loss.py:65-95)."""
    gv, gh = _corr3(x, "v"), _corr3(x, "h")
    if normal is None:
        return torch.sqrt(gv * gv + gh * gh + 1e-6)
    glr, grl = _corr3(x, "lr"), _corr3(x, "rl")
    code = direction_code(normal)
    g = gh.abs()
    g = torch.where(code == 1, gv.abs(), g)
    g = torch.where(code == 2, grl.abs(), g)
    g = torch.where(code == 3, glr.abs(), g)
    return g


def balanced_bce(edge, mask, prob, pos_to_neg=1.0):
    """comp_cross_entropy (grad_loss.py:161-219), edge_loss_class_list_to_mask_out == []."""
    m = torch.ones_like(edge) if mask is None else mask
    pos = -edge * torch.log(prob + 0.001)
    neg = -(1.0 - edge) * torch.log(1.0 - prob + 0.001)
    w_pos = (edge * m).sum(dim=(1, 2, 3))
    w_neg = ((1.0 - edge) * m).sum(dim=(1, 2, 3))
    if float(w_neg.sum()) == 0.0:
        alpha = torch.ones_like(w_neg)
    else:
        alpha = w_neg / (w_pos + w_neg)
    vals = torch.unique(m)
    if vals.numel() == 2 and bool((vals == 1).any()) and bool((vals == 0).any()):
        keep = (m != 0).to(edge.dtype)
        pos, neg = pos * keep, neg * keep
        n_valid = m.sum()
    else:
        n_valid = float(edge.numel())
    total = (pos_to_neg * alpha * pos.sum(dim=(1, 2, 3)) + (1.0 - alpha) * neg.sum(dim=(1, 2, 3))).sum()
    return total / n_valid


def grad_loss(output, gt_edge, gt_mask=None, is_grad=True, is_sigmoid=True, sigmoid_thresh=4.0,
              gt_normals=None, weight=10.0, pos_to_neg=1.0):
    """GradLoss.forward with edge_loss_type='cross_entropy' (grad_loss.py:122-159).
    Returns (loss, g.detach())."""
    if tuple(output.shape[-2:]) != tuple(gt_edge.shape[-2:]):
        output = F.interpolate(output, size=tuple(gt_edge.shape[-2:]), mode="bilinear")
    g = grad_layer(output, gt_normals) if is_grad else output
    p = torch.sigmoid(g - sigmoid_thresh) if is_sigmoid else g
    loss = weight * balanced_bce(gt_edge, gt_mask, p, pos_to_neg)
    return loss, g.detach()


def silog(pred, gt, ratio=10.0, ratio2=0.85):
    d = torch.log(pred * ratio) - torch.log(gt * ratio)
    return torch.sqrt((d * d).mean() - ratio2 * d.mean() ** 2) * ratio


def supervised_silog_loss(inv_depth0, gt_depth):
    """'sparse-silog', supervised_num_scales=1 (yaml:7-8): scale 0 only, valid = gt_inv > 0,
    +1e-5 on the prediction (supervised_loss.py:172-180)."""
    gt_inv = depth2inv(gt_depth)
    valid = gt_inv > 0.0
    return silog(inv_depth0[valid] + 1e-5, gt_inv[valid])


def edge_loss_all_scales(inv_depths, batch, mask=None, edge_weight=10.0, pos_to_neg=1.0):
    """compute_edge_loss_with_all_scales (SemiSupEdgeModel.py:164-198)."""
    total = 0.0
    for s in range(4):
        sfx = "" if s == 0 else "_%d" % s
        l, _ = grad_loss(inv2depth(inv_depths[s]), batch["edge" + sfx], mask, True, True, 4.0,
                         batch.get("normal" + sfx), weight=edge_weight, pos_to_neg=pos_to_neg)
        total = total + l
    return total / 4.0


def semisup_edge_model_loss(inv_depths, batch, supervised_loss_weight=1.0, model_edge_weight=1.0,
                            edge_weight=10.0, pos_to_neg=1.0):
    """SemiSupEdgeModel.forward training branch with supervised_loss_weight == 1
    (SemiSupEdgeModel.py:121-162).  Returns dict(loss[1], edge_loss, supervised_loss)."""
    mask = batch.get("rgb_edge")
    edge = edge_loss_all_scales(inv_depths, batch, mask, edge_weight, pos_to_neg)
    sup = supervised_loss_weight * supervised_silog_loss(inv_depths[0], batch["depth"]).unsqueeze(0)
    edge = model_edge_weight * edge
    loss = torch.zeros(1, dtype=inv_depths[0].dtype) + sup + edge
    return {"loss": loss, "edge_loss": edge.detach(), "supervised_loss": sup.detach()}


def flip_lr(x):
    return torch.flip(x, [3])


def adam_step(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (weight_decay=0, amsgrad=False) single-tensor update, step >= 1."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


def synthetic_batch(B, H, W, seed=0, dtype=torch.float32, with_mask=False):
    """SURVEY.md 8(d) synthetic inputs (host-side recipe; identical on CPU and GPU legs)."""
    g = torch.Generator()
    g.manual_seed(seed)
    batch = {"rgb": torch.rand(B, 3, H, W, generator=g)}
    dens = (torch.rand(B, 1, H, W, generator=g) < 0.05).float()
    batch["depth"] = dens * (1.0 + 79.0 * torch.rand(B, 1, H, W, generator=g))
    for s in range(4):
        sfx = "" if s == 0 else "_%d" % s
        h, w = H >> s, W >> s
        on = (torch.rand(B, 1, h, w, generator=g) < 0.03).float()
        batch["edge" + sfx] = on * torch.rand(B, 1, h, w, generator=g)
        batch["normal" + sfx] = (torch.rand(B, 1, h, w, generator=g) * 2 - 1) * math.pi
    if with_mask:
        batch["rgb_edge"] = (torch.rand(B, 1, H, W, generator=g) < 0.8).float()
    return {k: v.to(dtype) for k, v in batch.items()}
