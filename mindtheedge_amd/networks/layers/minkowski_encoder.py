"""Sparse auxiliary (SAN) branch of PackNet-SAN, inference and training -- SURVEY.md 8 row f-1.  **Parity unpinned.**

Mirror of packnet_sfm/networks/layers/minkowski_encoder.py (``MinkConv2D`` :11-86, ``MinkowskiEncoder`` :89-132) and of
``sparsify_depth`` / ``densify_features`` (networks/layers/minkowski.py:33-79) WITHOUT MinkowskiEngine: the sparse
operators are evaluated in their dense-equivalent form on zero-filled NHWC maps plus a byte mask of the active set
(csrc/san.hip), and the convolutions run on the library's dense MFMA kernels (a gather-GEMM-scatter form over the active sites exists
too, round 3: see SPARSE_GATHER below for why it is not the default).  MinkowskiEngine is a third-party CUDA
dependency of the reference that is not available where this was written, so the operator semantics follow its published
documentation as restated in oracle/san_oracle.py and cannot be checked against the real thing here:

  * MinkowskiConvolution(k, stride=1, bias=False): output on the input's active set, taps on the centred k x k window,
    kernel parameter ``kernel`` of shape [k*k, C_in, C_out]; tap order ASSUMED row-major over (row offset, column offset)
    in the coordinate order of sparsify_depth (v = row first, u = column second) -- this only matters when loading a
    checkpoint trained with MinkowskiEngine and must be confirmed against one before trusting such weights;
  * MinkowskiMaxPooling(3, stride=2): output cell (i, j) exists iff one of the fine cells (2i..2i+1, 2j..2j+1) does and
    takes the maximum over the active fine cells in rows 2i-1..2i+1, columns 2j-1..2j+1;
  * MinkowskiBatchNorm: BatchNorm1d over the active points of the whole batch (``bn.*`` keys): training mode normalises with
    the batch mean / biased variance of the active points and updates the running statistics (momentum 0.1, unbiased variance),
    eval mode uses the running statistics.

Parameter names match the reference's state dict (``mconvs.<level>.layer3.0.kernel``, ``...layer3.1.bn.weight`` ...).  Every
operator has a backward pass (round 2): the convolutions through the dense data- / weight-gradient kernels, the rest through
the masked kernels of csrc/san.hip (mte_sparse_bn_stats / _bn_relu_bwd / _maxpool3s2_bwd / mte_san_fuse_bwd), checked against
autograd through oracle/san_oracle.py.
"""
import torch
import torch.nn as nn

from ... import kernels as K


# Form of the branch's convolutions.  False (default): dense convolution of the zero-filled map, result kept on the mask.  True: gather-GEMM-
# scatter over the active sites (round 3, mte_conv2d_igemm_sparse).  Both are tested against the same statement and against each other
# (bit-identical on the active sites); the dense form is the default because it is FASTER on this workload -- measured on MI355X, B 4,
# 384x1280, 5 % LiDAR density (tools/san_bench.py, profiles/r03_san_dense_vs_gather.txt): branch 2.30 ms dense vs 2.89 ms gather.  Only the
# INPUT is 5 % active: the stride-2 3x3 pooling in front of every level leaves 18.5 % of the cells active at the first level, 55.7 % at
# the second and >= 96 % from the third on, and a gathered 5x5 site re-reads 25 scattered taps where the dense kernel streams every pixel once.
SPARSE_GATHER = False


class _SparseConvFn(torch.autograd.Function):
    """MinkowskiConvolution(k, stride 1) of the zero-filled map with the [k*k, C_in, C_out] kernel parameter.
    Round 3: gather-GEMM-scatter over the ACTIVE sites (`sites`: kernels.SiteList of the level's mask) on the implicit-GEMM MFMA kernel --
    a site's k x k taps are gathered from the zero-filled dense map, the result is scattered to the same sites and nothing else is
    written (every consumer selects by the mask), so the work follows the active count (~5 % of the pixels at the first level) instead
    of the dense map; forward and data gradient.  The weight gradient stays the dense reduction (inactive sites contribute exact zeros).
    sites = None: the dense-equivalent form of round 2 (the caller masks the result)."""

    @staticmethod
    def forward(ctx, feat, kernel, k, pack, sites=None):
        cin, cout = kernel.shape[1], kernel.shape[2]
        w = kernel.detach().view(k, k, cin, cout).permute(3, 2, 0, 1).contiguous()
        pack.key = None
        wf, _ = pack.get(w, feat.dtype, bool(ctx.needs_input_grad[0]))
        y = K.conv_forward(feat, wf, None, cout, k, k, pack=pack, w=w, sites=sites)
        ctx.save_for_backward(feat, w)
        ctx.pack, ctx.k, ctx.sites = pack, k, sites
        return y

    @staticmethod
    def backward(ctx, dy):
        feat, w = ctx.saved_tensors
        dy = K.as_act(dy, feat.dtype)
        dx, dw, _ = K.conv_backward(feat, dy, w, ctx.pack, ctx.needs_input_grad[0], need_dbias=False, sites=ctx.sites)
        k = ctx.k
        dkernel = None if dw is None else dw.permute(2, 3, 1, 0).reshape(k * k, w.shape[1], w.shape[0])
        return dx, dkernel, None, None, None


class _SparseConv(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        kernel = torch.empty(k * k, cin, cout)
        nn.init.kaiming_uniform_(kernel.view(k * k * cin, cout).t(), a=5 ** 0.5)
        self.kernel = nn.Parameter(kernel)
        self._pack = K.WeightPack()

    def forward(self, feat, sites=None):
        return _SparseConvFn.apply(feat, self.kernel, self.k, self._pack, sites)


class _SparseBatchNorm(nn.Module):
    """MinkowskiBatchNorm: the parameters live under ``.bn`` like the reference's wrapped BatchNorm1d."""

    def __init__(self, c):
        super().__init__()
        self.bn = nn.BatchNorm1d(c)


def _ptrs(*ts):
    out = []
    for t in ts:
        out += list(K._pl(t)) if t is not None else [None, 0]
    return out


class _SparseBnReluFn(torch.autograd.Function):
    """out = mask ? relu(bn(a [+ b] [+ c])) : 0 with the statistics handed in (batch statistics while training)."""

    @staticmethod
    def forward(ctx, a, b, c, mask, gamma, beta, mean, var, eps, n_active):
        B, C, H, W = a.shape
        out = K.new_act(B, C, H, W, a.dtype, a.device)
        po, lo = K._pl(out)
        K.lib.mte_sparse_bn_relu(*_ptrs(a, b, c), mask.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(),
                                 float(eps), po, lo, B * H * W, C, K._dt(a), K._stream())
        ctx.save_for_backward(a, b, c, mask, gamma, mean, var, out, n_active)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b, c, mask, gamma, mean, var, out, n_active = ctx.saved_tensors
        if n_active is None:
            raise K.MteError("the sparse BatchNorm backward needs batch statistics: put the module in train() mode")
        B, C, H, W = a.shape
        dout = K.as_act(dout, a.dtype)
        invstd = torch.rsqrt(var + ctx.eps)
        sums = torch.empty(2 * C, dtype=torch.float64, device=a.device)
        dx = K.new_act(B, C, H, W, a.dtype, a.device)
        px, lx = K._pl(dx)
        K.lib.mte_sparse_bn_relu_bwd(*_ptrs(a, b, c, out, dout), mask.data_ptr(), gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                     n_active.data_ptr(), sums.data_ptr(), px, lx, B * H * W, C, K._dt(a), K._stream())
        dbeta, dgamma = sums[:C].float(), sums[C:].float()
        return dx, (dx if b is not None else None), (dx if c is not None else None), None, dgamma, dbeta, None, None, None, None


def _bn_relu(a, mask, norm, b=None, c=None):
    bn = norm.bn
    if not norm.training:
        return _SparseBnReluFn.apply(a, b, c, mask, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, None)
    # MinkowskiBatchNorm in training mode: statistics over the active points of the whole batch
    B, C, H, W = a.shape
    sums = torch.empty(2 * C + 1, dtype=torch.float64, device=a.device)
    K.lib.mte_sparse_bn_stats(*_ptrs(a, b, c), mask.data_ptr(), sums.data_ptr(), B * H * W, C, K._dt(a), K._stream())
    n = sums[2 * C:].clamp(min=1.0)
    mean64 = sums[:C] / n
    var64 = (sums[C:2 * C] / n - mean64 * mean64).clamp(min=0.0)          # biased, as BatchNorm normalises
    mean, var = mean64.float(), var64.float()
    with torch.no_grad():
        m = bn.momentum if bn.momentum is not None else 0.1
        bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
        bn.running_var.mul_(1 - m).add_((var64 * (n / (n - 1).clamp(min=1.0))).float(), alpha=m)
        bn.num_batches_tracked += 1
    return _SparseBnReluFn.apply(a, b, c, mask, bn.weight, bn.bias, mean, var, bn.eps, sums[2 * C:])


class _SparseMaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, mask):
        B, C, H, W = feat.shape
        pooled = K.new_act(B, C, H // 2, W // 2, feat.dtype, feat.device)
        mask2 = torch.empty((B, H // 2, W // 2), dtype=torch.uint8, device=feat.device)
        pi, li = K._pl(feat)
        po, lo = K._pl(pooled)
        K.lib.mte_sparse_maxpool3s2(pi, li, mask.data_ptr(), po, lo, mask2.data_ptr(), B, H, W, C, K._dt(feat), K._stream())
        ctx.save_for_backward(feat, mask)
        ctx.mark_non_differentiable(mask2)
        return pooled, mask2

    @staticmethod
    def backward(ctx, dpooled, _dmask):
        feat, mask = ctx.saved_tensors
        B, C, H, W = feat.shape
        dpooled = K.as_act(dpooled, feat.dtype)
        din = K.new_act(B, C, H, W, feat.dtype, feat.device)
        K.lib.mte_sparse_maxpool3s2_bwd(*_ptrs(feat), mask.data_ptr(), *_ptrs(dpooled, din), B, H, W, C, K._dt(feat), K._stream())
        return din, None


class MinkConv2D(nn.Module):
    """Three parallel sparse conv stacks (3, 2 and 1 convolutions deep) summed, BatchNorm + ReLU; stride-2 max pooling in
    front (reference minkowski_encoder.py:11-86).  ``with_uncertainty`` / ``add_rgb`` are not used by PackNetSAN01."""

    def __init__(self, in_planes, out_planes, kernel_size, stride, with_uncertainty=False, add_rgb=False):
        super().__init__()
        if with_uncertainty or add_rgb:
            raise NotImplementedError("with_uncertainty / add_rgb are never enabled by PackNetSAN01")
        k, o = kernel_size, out_planes
        self.layer3 = nn.Sequential(_SparseConv(in_planes, 2 * o, k), _SparseBatchNorm(2 * o), nn.Identity(),
                                    _SparseConv(2 * o, 2 * o, k), _SparseBatchNorm(2 * o), nn.Identity(),
                                    _SparseConv(2 * o, o, k))
        self.layer2 = nn.Sequential(_SparseConv(in_planes, 2 * o, k), _SparseBatchNorm(2 * o), nn.Identity(),
                                    _SparseConv(2 * o, o, k))
        self.layer1 = nn.Sequential(_SparseConv(in_planes, o, k))
        self.layer_final = nn.Sequential(_SparseBatchNorm(o), nn.Identity())
        self.stride = stride

    def forward(self, x):
        feat, mask = x
        if self.stride != 1:
            feat, mask = _SparseMaxPoolFn.apply(feat, mask)
        l3, l2 = self.layer3, self.layer2
        sites = K.SiteList(mask) if SPARSE_GATHER else None          # one site list per level, shared by its six convolutions
        x1 = self.layer1[0](feat, sites)
        x2 = l2[3](_bn_relu(l2[0](feat, sites), mask, l2[1]), sites)
        x3 = l3[6](_bn_relu(l3[3](_bn_relu(l3[0](feat, sites), mask, l3[1]), sites), mask, l3[4]), sites)
        return None, (_bn_relu(x1, mask, self.layer_final[0], x2, x3), mask)


class MinkowskiEncoder(nn.Module):
    """prep(depth) then one call per pyramid level, each returning the densified features of that level
    (reference minkowski_encoder.py:89-132).  The dense map IS the densified tensor, so ``densify_features`` is free."""

    def __init__(self, channels, with_uncertainty=False, add_rgb=False):
        super().__init__()
        if with_uncertainty or add_rgb:
            raise NotImplementedError("with_uncertainty / add_rgb are never enabled by PackNetSAN01")
        ks = [5, 5] + [3] * (len(channels) - 1)
        self.mconvs = nn.ModuleList([MinkConv2D(1, channels[0], ks[0], 2)])
        for i in range(len(channels) - 1):
            self.mconvs.append(MinkConv2D(channels[i], channels[i + 1], ks[i + 1], 2))
        self.d, self.n, self.shape = None, 0, None

    def prep(self, d):
        """sparsify_depth (minkowski.py:33-57): active set = pixels with depth > 0, feature = the depth value."""
        K._require_gpu(d)
        B, C, H, W = d.shape
        if C != 1 or H % 32 or W % 32:
            raise ValueError("input_depth must be [B,1,H,W] with H and W multiples of 32, got {}".format(tuple(d.shape)))
        feat = K.new_act(B, 8, H, W, K.compute_dtype(), d.device)
        mask = torch.empty((B, H, W), dtype=torch.uint8, device=d.device)
        pf, lf = K._pl(feat)
        K.lib.mte_sparsify_depth(d.detach().float().contiguous().data_ptr(), pf, lf, mask.data_ptr(), B, H, W, K._dt(feat), K._stream())
        self.d, self.shape, self.n = (feat, mask), d.shape, 0

    def forward(self, x=None):
        _, self.d = self.mconvs[self.n](self.d)
        self.n += 1
        return self.d[0]


class _SanFuseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skip, sparse, weight, bias, index):
        B, C, H, W = skip.shape
        out = K.new_act(B, C, H, W, skip.dtype, skip.device)
        po, lo = K._pl(out)
        w1 = weight.detach()[index:index + 1].contiguous()
        b1 = bias.detach()[index:index + 1].contiguous()
        K.lib.mte_san_fuse(*_ptrs(skip, sparse), w1.data_ptr(), b1.data_ptr(), po, lo, B * H * W, C, K._dt(skip), K._stream())
        ctx.save_for_backward(skip, w1)
        ctx.index, ctx.n = index, weight.numel()
        return out

    @staticmethod
    def backward(ctx, dout):
        skip, w1 = ctx.saved_tensors
        B, C, H, W = skip.shape
        dout = K.as_act(dout, skip.dtype)
        dskip = K.new_act(B, C, H, W, skip.dtype, skip.device)
        sums = torch.empty(2, dtype=torch.float64, device=skip.device)
        K.lib.mte_san_fuse_bwd(*_ptrs(skip, dout), w1.data_ptr(), *K._pl(dskip), sums.data_ptr(), B * H * W, C, K._dt(skip), K._stream())
        dw = torch.zeros(ctx.n, dtype=torch.float32, device=skip.device)
        db = torch.zeros(ctx.n, dtype=torch.float32, device=skip.device)
        dw[ctx.index] = sums[0].float()
        db[ctx.index] = sums[1].float()
        return dskip, dout, dw, db, None


def san_fuse(skip, sparse, weight, bias, index):
    """skip * weight[index] + sparse + bias[index] (PackNetSAN01.py:254-258) in one pass."""
    return _SanFuseFn.apply(skip, sparse, weight, bias, index)


class _FeatL2Fn(torch.autograd.Function):
    """mean((a.detach() - b)^2) over the logical elements (PackNetSAN01.py:340-342: a = RGB+LiDAR feature, b = RGB feature)"""

    @staticmethod
    def forward(ctx, a, b):
        B, C, H, W = b.shape
        s = torch.empty(1, dtype=torch.float64, device=b.device)
        K.lib.mte_feat_l2(*_ptrs(a, b), s.data_ptr(), None, 0, None, 0.0, B * H * W, C, K._dt(b), K._stream())
        ctx.save_for_backward(a, b)
        ctx.inv_n = 1.0 / (B * H * W * C)
        return (s * ctx.inv_n).float()

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        B, C, H, W = b.shape
        db = K.new_act(B, C, H, W, b.dtype, b.device)
        gs = g.detach().float().contiguous()
        K.lib.mte_feat_l2(*_ptrs(a, b), None, *K._pl(db), gs.data_ptr(), ctx.inv_n, B * H * W, C, K._dt(b), K._stream())
        return None, db


def feature_l2(a, b):
    """mean squared difference between two NHWC activations of equal logical shape; gradient flows into ``b`` only"""
    a, b = K.as_act(a.detach(), K.compute_dtype()), K.as_act(b, K.compute_dtype())
    if tuple(a.shape) != tuple(b.shape):
        raise ValueError("feature shapes differ: {} vs {}".format(tuple(a.shape), tuple(b.shape)))
    if b.shape[1] % 8:
        raise K.MteError("feature_l2 expects channel counts that are multiples of 8 (no channel padding inside the mean)")
    return _FeatL2Fn.apply(a, b)
