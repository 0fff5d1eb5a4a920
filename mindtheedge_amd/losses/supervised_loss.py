"""Supervised (silog) loss on a fused gfx950 reduction -- drop-in for the 'sparse-silog' configuration of
packnet_sfm/losses/supervised_loss.py (SupervisedLoss.forward :183-216, calculate_loss :155-180, SilogLoss :57-69).
"""
import torch
import torch.nn as nn

from .. import kernels as K


class LossBase(nn.Module):
    def __init__(self):
        super().__init__()
        self._logs = {}
        self._metrics = {}

    @property
    def logs(self):
        return self._logs

    @property
    def metrics(self):
        return self._metrics

    def add_metric(self, key, val):
        self._metrics[key] = val.detach()


class SupervisedLoss(LossBase):
    def __init__(self, supervised_method='sparse-l1', supervised_num_scales=4, progressive_scaling=0.0, **kwargs):
        super().__init__()
        if supervised_method != 'sparse-silog':
            if not any(supervised_method.endswith(s) for s in ('l1', 'mse', 'berhu', 'silog', 'abs_rel')):
                raise ValueError('Unknown supervised loss {}'.format(supervised_method))
            raise NotImplementedError("only 'sparse-silog' (the shipped training YAML) is built, got %r" % supervised_method)
        if progressive_scaling > 0.0:
            raise NotImplementedError("progressive scaling is disabled in every shipped config")
        self.supervised_method = supervised_method
        self.n = supervised_num_scales

    @property
    def logs(self):
        return {'supervised_num_scales': self.n}

    def forward(self, inv_depths, gt_depth, return_logs=False, progress=0.0, gt_is_inverse=False):
        """inv_depths: list of predicted inverse-depth maps; gt_depth: metric depth [B,1,H,W] with 0 = invalid.
        (The reference receives depth2inv(depth); pass gt_is_inverse=True for that calling convention --
        the valid set {gt_inv > 0} == {depth > 0} is identical.)  Unlike the reference, the caller's list is
        NOT mutated (upstream replaces inv_depths[0] by its masked 1-D gather, supervised_loss.py:175)."""
        if self.n != 1:
            raise NotImplementedError("supervised_num_scales=%d: the shipped config uses 1" % self.n)
        gt = gt_depth
        if gt_is_inverse:
            gt = torch.where(gt_depth > 0, 1.0 / gt_depth.clamp(min=1e-30), torch.zeros_like(gt_depth))
        if tuple(gt.shape[-2:]) != tuple(inv_depths[0].shape[-2:]):
            raise NotImplementedError("ground truth must be at the resolution of scale 0")
        loss = K.SilogFn.apply(inv_depths[0], gt)
        self.add_metric('supervised_loss', loss)
        return {'loss': loss.unsqueeze(0), 'metrics': self.metrics}
