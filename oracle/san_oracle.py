"""CPU statement of the sparse auxiliary (SAN) branch -- TEST INFRASTRUCTURE ONLY.  **Parity unpinned.**

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(mindtheedge_amd/) never does.  SURVEY.md 8 row f-1.

The reference evaluates this branch with MinkowskiEngine (pinned nowhere; 0.5.x API), a third-party CUDA library that is
absent from /root/reference and from this image, so no fixture can be generated and nothing below is checked against it.
What is restated here is MinkowskiEngine's published operator semantics in dense form, applied to the reference's call
sites (packnet_sfm/networks/layers/minkowski_encoder.py:11-132, minkowski.py:33-79, networks/depth/PackNetSAN01.py:248-258):

  sparse tensor         = (dense feature map that is zero off the active set, boolean mask of the active set)
  sparsify_depth        : mask = depth > 0, feature = depth
  MinkowskiMaxPooling(3, stride 2, dimension 2)
                        : coarse cell (i, j) is active iff a fine cell in rows 2i..2i+1, columns 2j..2j+1 is; its value is
                          the maximum over the ACTIVE fine cells of rows 2i-1..2i+1, columns 2j-1..2j+1
  MinkowskiConvolution(k, stride 1)
                        : no bias; y[p] = sum over the taps t of the centred k x k window of W[t] x[p + t] for active p, with
                          inactive neighbours contributing nothing; ``kernel`` is [k*k, C_in, C_out], tap t = (row, column)
                          offset in row-major order (ASSUMED, see mindtheedge_amd/networks/layers/minkowski_encoder.py)
  MinkowskiBatchNorm    : BatchNorm1d over the active points; eval mode = per-channel affine with the running statistics,
                          training mode = batch statistics (biased variance) of the active points of the whole batch, running
                          statistics updated with momentum 0.1 and the unbiased variance (torch.nn.functional.batch_norm on the
                          gathered [N_active, C] matrix).  Every function below is differentiable by torch autograd, which is what
                          the backward kernels of csrc/san.hip are checked against.
  MinkowskiReLU, x1 + x2 + x3 (same coordinate map), densify_features = zeros off the active set
"""
import torch
import torch.nn.functional as F


def sparsify_depth(depth):
    mask = depth > 0
    return depth * mask, mask


def max_pool(feat, mask):
    B, C, H, W = feat.shape
    m2 = F.max_pool2d(mask.float(), 2, 2) > 0
    neg = torch.where(mask, feat, torch.full_like(feat, float("-inf")))
    pooled = F.max_pool2d(F.pad(neg, (1, 0, 1, 0), value=float("-inf")), 3, 2)          # rows 2i-1..2i+1, columns 2j-1..2j+1
    return torch.where(m2, pooled, torch.zeros_like(pooled)), m2


def conv(feat, kernel, k):
    """kernel: [k*k, C_in, C_out] -> dense convolution of the zero-filled map (values off the mask are discarded later)."""
    cin, cout = kernel.shape[1], kernel.shape[2]
    w = kernel.view(k, k, cin, cout).permute(3, 2, 0, 1)
    return F.conv2d(feat, w, None, padding=k // 2)


def bn_relu(x, mask, bn):
    """bn: dict with weight, bias, running_mean, running_var, eps"""
    s = bn["weight"] / torch.sqrt(bn["running_var"] + bn["eps"])
    y = (x - bn["running_mean"].view(1, -1, 1, 1)) * s.view(1, -1, 1, 1) + bn["bias"].view(1, -1, 1, 1)
    return torch.relu(y) * mask


def bn_relu_train(x, mask, bn):
    """training mode: F.batch_norm over the gathered active points (updates bn['running_mean'/'running_var'] in place)"""
    on = mask[:, 0]
    pts = x.permute(0, 2, 3, 1)[on]                                  # [N_active, C]
    y = F.batch_norm(pts, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], True, 0.1, bn["eps"])
    out = torch.zeros_like(x).permute(0, 2, 3, 1).contiguous()
    out[on] = torch.relu(y)
    return out.permute(0, 3, 1, 2)


def _bn(P, prefix, eps=1e-5):
    return {"weight": P[prefix + ".bn.weight"], "bias": P[prefix + ".bn.bias"], "running_mean": P[prefix + ".bn.running_mean"],
            "running_var": P[prefix + ".bn.running_var"], "eps": eps}


def mink_conv2d(P, prefix, feat, mask, k, train=False):
    """One MinkConv2D level (stride 2): P is the state dict, prefix e.g. 'mconvs.mconvs.0'."""
    bn_relu = bn_relu_train if train else globals()["bn_relu"]
    feat, mask = max_pool(feat, mask)
    x1 = conv(feat, P[prefix + ".layer1.0.kernel"], k)
    x2 = conv(bn_relu(conv(feat, P[prefix + ".layer2.0.kernel"], k), mask, _bn(P, prefix + ".layer2.1")), P[prefix + ".layer2.3.kernel"], k)
    t = bn_relu(conv(feat, P[prefix + ".layer3.0.kernel"], k), mask, _bn(P, prefix + ".layer3.1"))
    t = bn_relu(conv(t, P[prefix + ".layer3.3.kernel"], k), mask, _bn(P, prefix + ".layer3.4"))
    x3 = conv(t, P[prefix + ".layer3.6.kernel"], k)
    return bn_relu(x1 + x2 + x3, mask, _bn(P, prefix + ".layer_final.0")), mask


def san_features(P, depth, prefix="mconvs.mconvs", train=False, levels=5):
    """-> the densified feature maps (levels H/2 .. H/32) of MinkowskiEncoder for ``depth`` [B,1,H,W]."""
    feat, mask = sparsify_depth(depth)
    out = []
    for level, k in enumerate([5, 5, 3, 3, 3][:levels]):
        feat, mask = mink_conv2d(P, "%s.%d" % (prefix, level), feat, mask, k, train=train)
        out.append(feat)
    return out
