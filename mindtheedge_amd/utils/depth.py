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


# ---- validation metrics on device (reference utils/depth.py:202-361) ---------------------------

_FUSE_METHODS = {'mean': 0, 'max': 1, 'min': 2}
_SCALE_FNS = {'resize': 0, 'top-center': 1}


def _f32_maps(*tensors):
    from .. import kernels as K
    out = []
    for t in tensors:
        K._require_gpu(t)
        if t.dim() != 4 or t.shape[1] != 1:
            raise ValueError("expected a [B,1,H,W] map, got {}".format(tuple(t.shape)))
        out.append(t.detach().float().contiguous())
    return out


def fuse_inv_depth(inv_depth, inv_depth_hat, method='mean'):
    """Reference utils/depth.py:202-227 (tiny elementwise op; the fused kernel below does not call it)."""
    if method == 'mean':
        return 0.5 * (inv_depth + inv_depth_hat)
    if method == 'max':
        return torch.max(inv_depth, inv_depth_hat)
    if method == 'min':
        return torch.min(inv_depth, inv_depth_hat)
    raise ValueError('Unknown post-process method {}'.format(method))


def post_process_inv_depth(inv_depth, inv_depth_flipped, method='mean'):
    """Flip-TTA fusion of an inverse depth map with the prediction on the mirrored image (reference
    utils/depth.py:230-256) as one kernel: un-flip, fuse, and blend the 5 % border ramps."""
    from .. import kernels as K
    if method not in _FUSE_METHODS:
        raise ValueError('Unknown post-process method {}'.format(method))
    a, f = _f32_maps(inv_depth, inv_depth_flipped)
    if a.shape != f.shape:
        raise ValueError("shape mismatch {} vs {}".format(tuple(a.shape), tuple(f.shape)))
    out = torch.empty_like(a)
    B, _, H, W = a.shape
    K.lib.mte_post_process_inv_depth(a.data_ptr(), f.data_ptr(), out.data_ptr(), B, H, W, _FUSE_METHODS[method], K._stream())
    return out


def compute_depth_metrics(config, gt, pred, use_gt_scale=True):
    """(abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3) of ``pred`` against ``gt``, averaged over the batch, as a
    float32[7] DEVICE tensor (reference utils/depth.py:259-325; ``config`` carries crop / min_depth / max_depth /
    scale_output as there).  No host synchronisation: valid-pixel selection, scale_depth, the two medians and the
    reductions all run inside ``mte_depth_metrics``."""
    from .. import kernels as K
    scale_fn = getattr(config, 'scale_output', 'resize')
    if scale_fn not in _SCALE_FNS:
        raise NotImplementedError('Depth scale function {} not implemented.'.format(scale_fn))
    g, p = _f32_maps(gt, pred)
    B, _, H, W = g.shape
    if p.shape[0] != B:
        raise ValueError("batch mismatch {} vs {}".format(B, p.shape[0]))
    h, w = p.shape[-2:]
    nbytes = K.lib.mte_depth_metrics_workspace_bytes(B)
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=g.device)
    out = torch.empty(7, dtype=torch.float32, device=g.device)
    K.lib.mte_depth_metrics(g.data_ptr(), p.data_ptr(), B, H, W, h, w, _SCALE_FNS[scale_fn], int(getattr(config, 'crop', '') == 'garg'),
                            float(config.min_depth), float(config.max_depth), int(bool(use_gt_scale)), ws.data_ptr(), nbytes,
                            out.data_ptr(), K._stream())
    return out
