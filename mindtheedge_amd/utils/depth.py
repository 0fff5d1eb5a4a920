"""inv2depth / depth2inv (reference packnet_sfm/utils/depth.py:104-144).  The training path fuses both into the loss
kernels; these helpers serve inference post-processing (H3) on the small fp32 output maps."""
import torch


def inv2depth(inv_depth):
    if isinstance(inv_depth, (list, tuple)):
        return [inv2depth(i) for i in inv_depth]
    return 1. / inv_depth.clamp(min=1e-6)


def depth2inv(depth):
    if isinstance(depth, (list, tuple)):
        return [depth2inv(d) for d in depth]
    inv = 1. / depth.clamp(min=1e-6)
    inv[depth <= 0.] = 0.
    return inv
