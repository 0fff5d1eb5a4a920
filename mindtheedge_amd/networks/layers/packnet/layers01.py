"""PackNet layers on hand-written gfx950 kernels -- same class names, constructor arguments, parameter
names and OIHW fp32 parameter shapes as the reference (packnet_sfm/networks/layers/packnet/layers01.py), so
reference / TRI checkpoints load key-for-key.  Forward and backward run through mindtheedge_amd.kernels
(libmte_hip.so); nothing here falls back to torch.nn.functional.

Layers accept either a plain fp32 NCHW tensor (converted once to the NHWC compute-dtype layout) or an
activation already in that layout, and return NHWC activations of logical shape [B,C,H,W].
"""
import math

import torch
import torch.nn as nn

from .... import kernels as K


class _ConvParams(nn.Module):
    """Parameter holder with nn.Conv2d's names, shapes and default init (weight OIHW, bias)."""

    def __init__(self, in_channels, out_channels, kernel_size):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_channels * kernel_size * kernel_size)
        nn.init.uniform_(self.bias, -bound, bound)
        self.pack = K.WeightPack()


class _Conv3dParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d, 1, 3, 3, 3))
        self.bias = nn.Parameter(torch.empty(d))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.uniform_(self.bias, -1.0 / math.sqrt(27), 1.0 / math.sqrt(27))


class _GroupNormParams(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))


class _Dropout2dSpec(nn.Module):
    """Keeps the reference's nn.Sequential(conv, Dropout2d) key layout ('conv3.0.weight'); the mask itself is
    applied inside the fused residual-tail kernel as a per-(sample, channel) scale."""

    def __init__(self, p):
        super().__init__()
        self.p = p


def _enter(x, channels):
    """plain NCHW fp32 -> NHWC activation (zero channel padding to a multiple of 8); activations pass through."""
    if x.dtype == K.compute_dtype() and K.is_act(x) and x.shape[1] == K.round8(channels):
        return x
    if x.shape[1] == channels and x.dtype == torch.float32:
        return K.image_to_act(x)                     # (makes a strided image contiguous first)
    if x.shape[1] == K.round8(channels):
        return K.as_act(x, K.compute_dtype())
    raise K.MteError("expected %d (or %d padded) input channels, got %s" % (channels, K.round8(channels), tuple(x.shape)))


class Conv2D(nn.Module):
    """ELU(GroupNorm(16)(conv_k(zero_pad(x)))) -- reference layers01.py:11-38."""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        if stride != 1:
            raise NotImplementedError("the PackNetSAN01 path only uses stride-1 convolutions")
        self.kernel_size = kernel_size
        self.conv_base = _ConvParams(in_channels, out_channels, kernel_size)
        self.normalize = _GroupNormParams(out_channels)

    def forward(self, x, out=None, inv=None):
        """`out`: optional destination (e.g. K.channel_slice of a decoder concat buffer) for the result.  `inv`: the LAST input channel given as a
        half-resolution one-channel map to be nearest-up-sampled (decoder iconv layers): x then holds the other in_channels - 1 channels"""
        if inv is not None:
            x = _enter(x, self.conv_base.in_channels - 1)
            return K.ConvGnEluInvFn.apply(x, inv, self.conv_base.weight, self.conv_base.bias, self.normalize.weight, self.normalize.bias,
                                          self.conv_base.pack, out)
        x = _enter(x, self.conv_base.in_channels)
        return K.ConvGnEluFn.apply(x, self.conv_base.weight, self.conv_base.bias, self.normalize.weight, self.normalize.bias,
                                   self.conv_base.pack, out)


class ResidualConv(nn.Module):
    """ELU(GN(conv2(conv1(x)) + Dropout2d(conv1x1(x)))) -- reference layers01.py:41-73."""

    def __init__(self, in_channels, out_channels, stride, dropout=None):
        super().__init__()
        if stride != 1:
            raise NotImplementedError("stride-1 only")
        self.conv1 = Conv2D(in_channels, out_channels, 3, stride)
        self.conv2 = Conv2D(out_channels, out_channels, 3, 1)
        self.conv3 = _ConvParams(in_channels, out_channels, 1)
        self.normalize = _GroupNormParams(out_channels)
        self.dropout = dropout
        if dropout:
            self.conv3 = nn.Sequential(self.conv3, _Dropout2dSpec(dropout))

    def _shortcut_params(self):
        return self.conv3[0] if isinstance(self.conv3, nn.Sequential) else self.conv3

    def forward(self, x, channel_scale=None):
        sc = self._shortcut_params()
        x = _enter(x, sc.in_channels)
        xa, xb = K.fork(x)
        y1 = self.conv1(xa)
        # (its bias gradient comes out of the tail's backward pass)
        s = K.ConvFn.apply(xb, sc.weight, sc.bias, sc.pack, False)
        if channel_scale is None and self.dropout and self.training:
            channel_scale = K.dropout2d_scale(x.shape[0], sc.out_channels, self.dropout, x.device)
        B, _, H, W = x.shape
        if K.residual_tail_fused_ok(B, sc.out_channels, H, W, y1.dtype):
            # conv2 (a Conv2D) and the block's tail as one op: conv2's norm + ELU happen inside the kernel that forms the sum (kernels.ConvResidualTailFn)
            cb, gn = self.conv2.conv_base, self.conv2.normalize
            return K.ConvResidualTailFn.apply(y1, cb.weight, cb.bias, gn.weight, gn.bias, cb.pack, s, channel_scale,
                                              self.normalize.weight, self.normalize.bias, sc.bias)
        y = self.conv2(y1)
        return K.ResidualTailFn.apply(y, s, channel_scale, self.normalize.weight, self.normalize.bias, sc.bias)


def ResidualBlock(in_channels, out_channels, num_blocks, stride, dropout=None):
    layers = [ResidualConv(in_channels, out_channels, stride, dropout=dropout)]
    layers += [ResidualConv(out_channels, out_channels, 1, dropout=dropout) for _ in range(1, num_blocks)]
    return nn.Sequential(*layers)


class InvDepth(nn.Module):
    """sigmoid(conv3x3(x)) / min_depth -> fp32 [B,1,H,W] -- reference layers01.py:99-123."""

    def __init__(self, in_channels, out_channels=1, min_depth=0.5):
        super().__init__()
        if out_channels != 1:
            raise NotImplementedError("single-channel inverse depth only")
        self.min_depth = min_depth
        self.conv1 = _ConvParams(in_channels, out_channels, 3)

    def forward(self, x):
        x = _enter(x, self.conv1.in_channels)
        return K.InvDepthFn.apply(x, self.conv1.weight, self.conv1.bias, self.min_depth)


def packing(x, r=2):
    """Space-to-depth index map of the reference (layers01.py:127-149), as a pure view/permute for callers that
    want the packed tensor itself; the network path never materialises it (see Pack3dFn)."""
    b, c, h, w = x.shape
    return x.reshape(b, c, h // r, r, w // r, r).permute(0, 1, 3, 5, 2, 4).reshape(b, c * r * r, h // r, w // r)


class PackLayerConv3d(nn.Module):
    """packing -> Conv3d(1->d) -> view -> Conv2D -- reference layers01.py:214-248."""

    def __init__(self, in_channels, kernel_size, r=2, d=8):
        super().__init__()
        if r != 2 or d != 4:
            raise NotImplementedError("PackNetSAN01 uses r=2, d=4")
        self.in_channels = in_channels
        self.conv = Conv2D(in_channels * (r ** 2) * d, in_channels, kernel_size, 1)
        self.conv3d = _Conv3dParams(d)
        self.fold_pack = K.WeightPack()       # packs of the conv3d-folded (k+2)x(k+2) weights

    def forward(self, x, out=None):
        x = _enter(x, self.in_channels)
        k = self.conv.kernel_size
        if K.pack_folding_enabled() and self.in_channels % 8 == 0 and K.pack_fold_applicable(x.shape[2] // 2, x.shape[3] // 2, k):
            cb, gn = self.conv.conv_base, self.conv.normalize
            return K.PackFoldedConvGnEluFn.apply(x, self.conv3d.weight, self.conv3d.bias, cb.weight, cb.bias, gn.weight, gn.bias,
                                                 cb.pack, self.fold_pack, out)
        return self.conv(K.Pack3dFn.apply(x, self.conv3d.weight, self.conv3d.bias), out=out)


class UnpackLayerConv3d(nn.Module):
    """Conv2D -> Conv3d(1->d) -> view -> PixelShuffle(2) -- reference layers01.py:251-287."""

    def __init__(self, in_channels, out_channels, kernel_size, r=2, d=8):
        super().__init__()
        if r != 2 or d != 4:
            raise NotImplementedError("PackNetSAN01 uses r=2, d=4")
        self.conv = Conv2D(in_channels, out_channels * (r ** 2) // d, kernel_size, 1)
        self.conv3d = _Conv3dParams(d)

    def forward(self, x, out=None):
        return K.Unpack3dFn.apply(self.conv(x), self.conv3d.weight, self.conv3d.bias, out)
