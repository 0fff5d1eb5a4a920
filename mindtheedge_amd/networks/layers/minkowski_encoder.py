"""Sparse auxiliary (SAN) branch of PackNet-SAN, inference only -- SURVEY.md 8 row f-1.  **Parity unpinned.**

Mirror of packnet_sfm/networks/layers/minkowski_encoder.py (``MinkConv2D`` :11-86, ``MinkowskiEncoder`` :89-132) and of
``sparsify_depth`` / ``densify_features`` (networks/layers/minkowski.py:33-79) WITHOUT MinkowskiEngine: the sparse
operators are evaluated in their dense-equivalent form on zero-filled NHWC maps plus a byte mask of the active set
(csrc/san.hip), and the convolutions run on the library's dense MFMA kernels.  MinkowskiEngine is a third-party CUDA
dependency of the reference that is not available where this was written, so the operator semantics follow its published
documentation as restated in oracle/san_oracle.py and cannot be checked against the real thing here:

  * MinkowskiConvolution(k, stride=1, bias=False): output on the input's active set, taps on the centred k x k window,
    kernel parameter ``kernel`` of shape [k*k, C_in, C_out]; tap order ASSUMED row-major over (row offset, column offset)
    in the coordinate order of sparsify_depth (v = row first, u = column second) -- this only matters when loading a
    checkpoint trained with MinkowskiEngine and must be confirmed against one before trusting such weights;
  * MinkowskiMaxPooling(3, stride=2): output cell (i, j) exists iff one of the fine cells (2i..2i+1, 2j..2j+1) does and
    takes the maximum over the active fine cells in rows 2i-1..2i+1, columns 2j-1..2j+1;
  * MinkowskiBatchNorm: BatchNorm1d over the active points (``bn.*`` keys); only eval mode (running statistics) is built.

Parameter names match the reference's state dict (``mconvs.<level>.layer3.0.kernel``, ``...layer3.1.bn.weight`` ...), all
parameters are created with requires_grad=False (no backward pass exists for this branch).
"""
import torch
import torch.nn as nn

from ... import kernels as K


class _SparseConv(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        kernel = torch.empty(k * k, cin, cout)
        nn.init.kaiming_uniform_(kernel.view(k * k * cin, cout).t(), a=5 ** 0.5)
        self.kernel = nn.Parameter(kernel, requires_grad=False)
        self._pack, self._oihw, self._key = K.WeightPack(), None, None

    def oihw(self):
        key = (self.kernel.data_ptr(), self.kernel._version)
        if key != self._key:
            k = self.k
            self._oihw = self.kernel.detach().view(k, k, self.cin, self.cout).permute(3, 2, 0, 1).contiguous()
            self._key = key
        return self._oihw

    def forward(self, feat):
        """dense convolution of the zero-filled map (the caller masks the result)"""
        w = self.oihw()
        wf, _ = self._pack.get(w, feat.dtype, False)
        return K.conv_forward(feat, wf, None, self.cout, self.k, self.k, pack=self._pack, w=w)


class _SparseBatchNorm(nn.Module):
    """MinkowskiBatchNorm: the parameters live under ``.bn`` like the reference's wrapped BatchNorm1d."""

    def __init__(self, c):
        super().__init__()
        self.bn = nn.BatchNorm1d(c)
        for p in self.bn.parameters():
            p.requires_grad = False


def _bn_relu(a, mask, norm, b=None, c=None):
    if norm.training:
        raise NotImplementedError("the sparse branch is built for inference (eval mode) only: batch statistics over the active "
                                  "points and the backward pass are not implemented (SURVEY.md 8 f-1)")
    B, C, H, W = a.shape
    out = K.new_act(B, C, H, W, a.dtype, a.device)
    pa, la = K._pl(a)
    pb, lb = K._pl(b) if b is not None else (None, 0)
    pc, lc = K._pl(c) if c is not None else (None, 0)
    po, lo = K._pl(out)
    bn = norm.bn
    K.lib.mte_sparse_bn_relu(pa, la, pb, lb, pc, lc, mask.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
                             bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.eps), po, lo, B * H * W, C,
                             K._dt(a), K._stream())
    return out


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
            B, C, H, W = feat.shape
            pooled = K.new_act(B, C, H // 2, W // 2, feat.dtype, feat.device)
            mask2 = torch.empty((B, H // 2, W // 2), dtype=torch.uint8, device=feat.device)
            pi, li = K._pl(feat)
            po, lo = K._pl(pooled)
            K.lib.mte_sparse_maxpool3s2(pi, li, mask.data_ptr(), po, lo, mask2.data_ptr(), B, H, W, C, K._dt(feat), K._stream())
            feat, mask = pooled, mask2
        l3, l2 = self.layer3, self.layer2
        x1 = self.layer1[0](feat)
        x2 = l2[3](_bn_relu(l2[0](feat), mask, l2[1]))
        x3 = l3[6](_bn_relu(l3[3](_bn_relu(l3[0](feat), mask, l3[1])), mask, l3[4]))
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


def san_fuse(skip, sparse, weight, bias, index):
    """skip * weight[index] + sparse + bias[index] (PackNetSAN01.py:254-258) in one pass."""
    B, C, H, W = skip.shape
    out = K.new_act(B, C, H, W, skip.dtype, skip.device)
    ps, ls = K._pl(skip)
    pq, lq = K._pl(sparse)
    po, lo = K._pl(out)
    K.lib.mte_san_fuse(ps, ls, pq, lq, weight.detach()[index:index + 1].data_ptr(), bias.detach()[index:index + 1].data_ptr(),
                       po, lo, B * H * W, C, K._dt(skip), K._stream())
    return out
