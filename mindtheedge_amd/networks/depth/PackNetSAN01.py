"""PackNet-SAN depth network (dense RGB path) on gfx950 kernels -- drop-in for
packnet_sfm/networks/depth/PackNetSAN01.py: same constructor signature, module tree and state-dict keys
(encoder.*, decoder.*, weight, bias), same forward contract:

  train: {'inv_depths': [4 x fp32 [B,1,H/2^s,W/2^s]]}                         (reference :319-322)
  eval : {'inv_depths': [[4 maps], [skip0..skip4, x5p]]}                       (reference :282-293)

  train with input_depth (with_san=True): + 'inv_depths_rgbd', 'depth_loss'     (reference :324-342, two passes)

The sparse LiDAR (SAN) branch is materialised on request (``with_san=True``, SURVEY.md 8(f-1), parity unpinned: the reference
runs it on MinkowskiEngine); without it ``input_depth`` raises NotImplementedError instead of being silently ignored.
"""
import os

import torch
import torch.nn as nn

from ... import kernels as K
from ..layers.packnet.layers01 import (PackLayerConv3d, UnpackLayerConv3d, Conv2D, ResidualBlock, InvDepth,
                                      _ConvParams, _Conv3dParams)


class PackNetSlimEnc01(nn.Module):
    def __init__(self, version, in_channels, ni, n1, n2, n3, n4, n5, pack_kernel, num_blocks, num_3d_feat, dropout):
        super().__init__()
        self.version = version
        self.in_channels = in_channels
        self.pre_calc = Conv2D(in_channels, ni, 5, 1)
        self.pack1 = PackLayerConv3d(n1, pack_kernel[0], d=num_3d_feat)
        self.pack2 = PackLayerConv3d(n2, pack_kernel[1], d=num_3d_feat)
        self.pack3 = PackLayerConv3d(n3, pack_kernel[2], d=num_3d_feat)
        self.pack4 = PackLayerConv3d(n4, pack_kernel[3], d=num_3d_feat)
        self.pack5 = PackLayerConv3d(n5, pack_kernel[4], d=num_3d_feat)
        self.conv1 = Conv2D(ni, n1, 7, 1)
        self.conv2 = ResidualBlock(n1, n2, num_blocks[0], 1, dropout=dropout)
        self.conv3 = ResidualBlock(n2, n3, num_blocks[1], 1, dropout=dropout)
        self.conv4 = ResidualBlock(n3, n4, num_blocks[2], 1, dropout=dropout)
        self.conv5 = ResidualBlock(n4, n5, num_blocks[3], 1, dropout=dropout)

    def forward(self, rgb, skip_out=None):
        """`skip_out`: optional destinations for the five skip tensors (channel blocks of the decoder's concat buffers,
        see Decoder.concat_buffers) so that torch.cat never has to copy them."""
        so = skip_out or [None] * 5
        x, s1 = K.fork(self.pre_calc(rgb, out=so[0]))                  # every skip has two consumers: next stage + decoder
        x1p, s2 = K.fork(self.pack1(self.conv1(x), out=so[1]))
        x2p, s3 = K.fork(self.pack2(self.conv2(x1p), out=so[2]))
        x3p, s4 = K.fork(self.pack3(self.conv3(x2p), out=so[3]))
        x4p, s5 = K.fork(self.pack4(self.conv4(x3p), out=so[4]))
        x5p = self.pack5(self.conv5(x4p))
        return x5p, [s1, s2, s3, s4, s5]


class Decoder(nn.Module):
    def __init__(self, version, out_channels, ni, n1, n2, n3, n4, n5, unpack_kernel, iconv_kernel, num_3d_feat):
        super().__init__()
        if version != 'A':
            raise NotImplementedError("only feature stacking 'A' (concatenation) is built; the shipped YAMLs use '1A'")
        self.version = version
        self.unpack5 = UnpackLayerConv3d(n5, n5, unpack_kernel[0], d=num_3d_feat)
        self.unpack4 = UnpackLayerConv3d(n5, n4, unpack_kernel[1], d=num_3d_feat)
        self.unpack3 = UnpackLayerConv3d(n4, n3, unpack_kernel[2], d=num_3d_feat)
        self.unpack2 = UnpackLayerConv3d(n3, n2, unpack_kernel[3], d=num_3d_feat)
        self.unpack1 = UnpackLayerConv3d(n2, n1, unpack_kernel[4], d=num_3d_feat)
        self.iconv5 = Conv2D(n5 + n4, n5, iconv_kernel[0], 1)
        self.iconv4 = Conv2D(n4 + n3, n4, iconv_kernel[1], 1)
        self.iconv3 = Conv2D(n3 + n2 + out_channels, n3, iconv_kernel[2], 1)
        self.iconv2 = Conv2D(n2 + n1 + out_channels, n2, iconv_kernel[3], 1)
        self.iconv1 = Conv2D(n1 + ni + out_channels, n1, iconv_kernel[4], 1)
        self.disp4_layer = InvDepth(n4, out_channels=out_channels)
        self.disp3_layer = InvDepth(n3, out_channels=out_channels)
        self.disp2_layer = InvDepth(n2, out_channels=out_channels)
        self.disp1_layer = InvDepth(n1, out_channels=out_channels)

    def concat_buffers(self, B, H, W, device):
        """The five torch.cat((unpack, skip[, up(inv_depth)]), 1) results of reference :118-143, allocated up front so the
        unpack layers and the encoder write their outputs straight into them.  -> (buffers level 1..5, skip destinations)"""
        bufs, skip_dst = [], []
        for lvl, (unp, ico) in enumerate(((self.unpack1, self.iconv1), (self.unpack2, self.iconv2), (self.unpack3, self.iconv3),
                                          (self.unpack4, self.iconv4), (self.unpack5, self.iconv5))):
            cu = unp.conv.conv_base.out_channels                       # unpack output channels (out * r^2 / d == out)
            with_inv = lvl < 3
            cs = ico.conv_base.in_channels - cu - (1 if with_inv else 0)
            buf = K.new_concat_buffer(B, [cu, cs], with_inv and not self._split_inv(cu + cs, ico.conv_base.out_channels, H >> lvl, W >> lvl),
                                      H >> lvl, W >> lvl, device=device)
            bufs.append(buf)
            skip_dst.append(K.channel_slice(buf, cu, cu + cs))
        return bufs, skip_dst

    @staticmethod
    def _split_inv(c_main, c_out, h, w):
        """the up-sampled inverse-depth input of iconv3 / iconv2 / iconv1 as a rank-1 term beside the GEMM (kernels.ConvGnEluInvFn) instead of a
        65th / 97th / 193rd channel of the concat buffer: when enabled, the other channels are a multiple of 32, and ALL THREE rank-1 kernels take the
        layer -- mte_rank1_conv_bwd_data / _bwd_weight cover 32, 64 or 128 output channels and even image sizes (round-4 advisor: the forward alone
        accepts up to 256, so a wider decoder would have failed in its backward pass instead of taking the concat form)"""
        return (K._cfg["split_inv_channel"] and c_main % 32 == 0 and c_out in (32, 64, 128) and h % 2 == 0 and w % 2 == 0
                and K.compute_dtype() in (torch.bfloat16, torch.float32))

    def _iconv(self, layer, inv, buf, up, skip):
        """iconv(cat(unpack output, skip[, nearest_up2(inv)]))"""
        if (inv is not None and self._split_inv(up.shape[1] + skip.shape[1], layer.conv_base.out_channels, up.shape[2], up.shape[3])
                and (buf is None or buf.shape[1] == up.shape[1] + skip.shape[1])):
            return layer(K.ConcatFn.apply(None, buf, up, skip), inv=inv)
        return layer(K.ConcatFn.apply(inv, buf, up, skip))

    def forward(self, x5p, skips, bufs=None):
        skip1, skip2, skip3, skip4, skip5 = skips
        b1, b2, b3, b4, b5 = bufs or [None] * 5

        def up_dst(buf, unp):
            return None if buf is None else K.channel_slice(buf, 0, unp.conv.conv_base.out_channels)

        iconv5 = self.iconv5(K.ConcatFn.apply(None, b5, self.unpack5(x5p, out=up_dst(b5, self.unpack5)), skip5))
        iconv4, f4 = K.fork(self.iconv4(K.ConcatFn.apply(None, b4, self.unpack4(iconv5, out=up_dst(b4, self.unpack4)), skip4)))
        inv_depth4 = self.disp4_layer(f4)
        iconv3, f3 = K.fork(self._iconv(self.iconv3, inv_depth4, b3, self.unpack3(iconv4, out=up_dst(b3, self.unpack3)), skip3))
        inv_depth3 = self.disp3_layer(f3)
        iconv2, f2 = K.fork(self._iconv(self.iconv2, inv_depth3, b2, self.unpack2(iconv3, out=up_dst(b2, self.unpack2)), skip2))
        inv_depth2 = self.disp2_layer(f2)
        iconv1 = self._iconv(self.iconv1, inv_depth2, b1, self.unpack1(iconv2, out=up_dst(b1, self.unpack1)), skip1)
        inv_depth1 = self.disp1_layer(iconv1)
        return [inv_depth1, inv_depth2, inv_depth3, inv_depth4]


class _SparseBranchPlaceholder(nn.Module):
    """Stands where the reference's MinkowskiEncoder ('mconvs') sits; holds no parameters."""


class PackNetSAN01(nn.Module):
    def __init__(self, dropout=None, version=None, freeze_encoder=False, freeze_decoder=False, freeze_san=False,
                 input_channels=3, is_depth_aux_net=False, output_channels=1, with_san=False, **kwargs):
        super().__init__()
        self.version = version[1:]
        self.in_channels = input_channels
        self.is_depth_aux_net = is_depth_aux_net      # the reference reads this attribute but never sets it
        if is_depth_aux_net:
            raise NotImplementedError("depth_aux_net does not exist in the reference either (never constructed)")
        ni, n1, n2, n3, n4, n5 = 32, 32, 64, 128, 256, 512
        num_blocks = [2, 2, 3, 3]
        pack_kernel = [5, 3, 3, 3, 3]
        unpack_kernel = [3, 3, 3, 3, 3]
        iconv_kernel = [3, 3, 3, 3, 3]
        num_3d_feat = 4
        self.encoder = PackNetSlimEnc01(self.version, self.in_channels, ni, n1, n2, n3, n4, n5,
                                        pack_kernel, num_blocks, num_3d_feat, dropout)
        if freeze_encoder:
            self.freeze_weights(self.encoder.parameters())
        self.decoder = Decoder(self.version, output_channels, ni, n1, n2, n3, n4, n5, unpack_kernel, iconv_kernel, num_3d_feat)
        if freeze_decoder:
            self.freeze_weights(self.decoder.parameters())
        # The reference always owns the sparse branch (34 M parameters) although SemiSupEdgeModel never runs it; here it is
        # materialised on request (with_san=True: inference or training with a LiDAR input) so that the training path's parameter set,
        # flat optimizer buffers and fixtures stay those of the 218 dense tensors.  PARITY UNPINNED, see minkowski_encoder.py.
        if with_san:
            from ..layers.minkowski_encoder import MinkowskiEncoder
            self.mconvs = MinkowskiEncoder([n1, n2, n3, n4, n5], with_uncertainty=False)
        else:
            self.mconvs = _SparseBranchPlaceholder()
        self.with_san = bool(with_san)
        self.weight = nn.Parameter(torch.ones(5), requires_grad=not freeze_san)
        self.bias = nn.Parameter(torch.zeros(5), requires_grad=not freeze_san)
        self.weight._mte_flat_tail = self.bias._mte_flat_tail = True     # used only with input_depth: see trainers/data_parallel.py
        self.init_weights()

    def init_weights(self):
        """xavier-uniform conv / conv3d weights, zero biases (reference :214-220)."""
        for m in self.modules():
            if isinstance(m, (_ConvParams, _Conv3dParams)):
                nn.init.xavier_uniform_(m.weight)
                m.bias.data.zero_()

    @staticmethod
    def freeze_weights(params):
        for p in params:
            p.requires_grad = False

    def run_network(self, rgb, input_depth=None):
        if input_depth is not None and not self.with_san:
            raise NotImplementedError("construct PackNetSAN01(with_san=True) to use input_depth: the sparse LiDAR (SAN) branch "
                                      "is only materialised on request (SURVEY.md 8 f-1, parity unpinned)")
        bufs = skip_dst = None
        if input_depth is None and rgb.is_cuda and rgb.shape[2] % 32 == 0 and rgb.shape[3] % 32 == 0 \
                and not os.environ.get("MTE_NO_CONCAT_PLACEMENT"):
            bufs, skip_dst = self.decoder.concat_buffers(rgb.shape[0], rgb.shape[2], rgb.shape[3], rgb.device)
        x5p, skips = self.encoder(rgb, skip_out=skip_dst)
        if input_depth is not None:
            # reference :248-258: every pyramid level below full resolution becomes skip * w + sparse features + b
            from ..layers.minkowski_encoder import san_fuse
            self.mconvs.prep(input_depth)
            skips = list(skips)
            for level in range(5):
                dense = skips[level + 1] if level < 4 else x5p
                sparse = self.mconvs(dense)
                if tuple(sparse.shape) != tuple(dense.shape):
                    raise RuntimeError("sparse level {} is {} but the encoder feature is {}".format(level, tuple(sparse.shape), tuple(dense.shape)))
                fused = san_fuse(dense, sparse, self.weight, self.bias, level)
                if level < 4:
                    skips[level + 1] = fused
                else:
                    x5p = fused
        return [self.decoder(x5p, skips, bufs), skips + [x5p]]

    def forward(self, rgb, input_depth=None, rgb_edge=None, output_features=False, **kwargs):
        if self.in_channels == 4:
            rgb = torch.cat((rgb, rgb_edge), dim=1)
        if not self.training:
            out = self.run_network(rgb, input_depth)
            if self.in_channels == 4:
                out[0] = out[0] * rgb_edge            # as the reference: list * tensor is an upstream bug; kept failing loudly
            return {'inv_depths': out}
        inv_depths, feats = self.run_network(rgb)
        output = {'inv_depths': inv_depths}
        if output_features:
            output['skip_feat_rgb'] = feats
        if input_depth is None:
            return output
        # reference :324-342: second pass with the LiDAR input through the sparse branch, and the feature-matching loss that
        # pulls the RGB-only pyramid towards the (detached) RGB+LiDAR one
        from ..layers.minkowski_encoder import feature_l2
        # the 216 encoder / decoder tensors are used by BOTH passes: their gradients must be summed, not stored twice into the
        # flat gradient buffer (and a bucket's all-reduce must not start on the first half) -- ordinary autograd accumulation
        # until the next zero_grad
        K.suspend_grad_sink()
        inv_depths_rgbd, feats_rgbd = self.run_network(rgb, input_depth)
        output['inv_depths_rgbd'] = inv_depths_rgbd
        loss = None
        for srgbd, srgb in zip(feats_rgbd, feats):
            term = feature_l2(srgbd, srgb)
            loss = term if loss is None else loss + term
        output['depth_loss'] = loss / len(feats_rgbd)
        if output_features:
            output['skip_feat_rgbd'] = feats_rgbd
        return output
