"""SemiSupEdgeModel: silog supervision + multi-scale depth-edge loss over the PackNet-SAN output -- drop-in
for packnet_sfm/models/SemiSupEdgeModel.py (forward :98-162, compute_edge_loss_with_all_scales :164-198).

loss = supervised_loss_weight * silog(inv_0, depth) + depth_edges_loss_weight * mean_s GradLoss(inv2depth(inv_s), edge_s, normal_s)

Differences from the reference, all behaviour-preserving for the shipped configuration:
  * 'input_depth' is not forwarded to the depth network: the RGB+LiDAR pass it triggers upstream never reaches
    the loss (only self_sup_output['inv_depths'] is consumed, reference :130,145).
  * inv2depth is fused into the edge-loss stencil (from_inv_depth=True), and the detached edge-strength maps the
    reference computes and drops are not written.
  * the caller's inv_depths list is not mutated by the supervised loss.
  * in the shipped configuration the four per-scale head calls and the silog loss run as ONE fused forward and ONE fused
    backward launch (``fuse_losses``); `edge_loss(...)` / `supervised_loss(...)` remain and give the same numbers per scale.
"""
import torch

from .SfmModel import SfmModel
from .model_utils import merge_outputs
from ..losses.supervised_loss import SupervisedLoss


class SemiSupEdgeModel(SfmModel):
    SIGMOID_THRESH = 4       # reference SemiSupEdgeModel.py:137-139: is_grad=True, is_sigmoid=True, sigmoid_thresh=4 (both loss paths use it)

    def __init__(self, supervised_loss_weight=0.9, depth_edges_loss_weight=10.0, **kwargs):
        super().__init__(**kwargs)
        assert 0. < supervised_loss_weight <= 1., "Model requires (0, 1] supervision"
        if supervised_loss_weight != 1.0:
            raise NotImplementedError("self-supervised photometric term (supervised_loss_weight < 1) is out of scope; "
                                      "the shipped YAML uses 1.0")
        self.supervised_loss_weight = supervised_loss_weight
        self._supervised_loss = SupervisedLoss(**kwargs)
        self._network_requirements.remove('pose_net')
        self._train_requirements.append('gt_depth')
        self.edges_depth_edge_loss_all_scales = kwargs['edges_depth_edge_loss_all_scales']
        self._input_keys = ['rgb', 'edge', 'rgb_edge', 'normal']
        if self.edges_depth_edge_loss_all_scales:
            self._input_keys += ['edge_1', 'edge_2', 'edge_3', 'normal_1', 'normal_2', 'normal_3']
        self.depth_edges_loss_weight = depth_edges_loss_weight
        self.fuse_losses = True          # one launch for every loss term (kernels.DepthLossesFn); False = one head call per scale

    @property
    def logs(self):
        return {**super().logs, **self._supervised_loss.logs}

    def supervised_loss(self, inv_depths, gt_depth, return_logs=False, progress=0.0):
        return self._supervised_loss(inv_depths, gt_depth, return_logs=return_logs, progress=progress)

    def edge_loss(self, pred, gt_edges, gt_mask=None, is_grad=True, is_sigmoid=True, sigmoid_thresh=4, gt_normals=None, **kw):
        return self.edge_loss_head(pred, gt_edges, gt_mask, is_grad, is_sigmoid, sigmoid_thresh, gt_normals, **kw)

    def compute_edge_loss_with_all_scales(self, inv_depths, batch, seg_mask, is_grad=False, is_sigmoid=False, sigmoid_thresh=4):
        total = None
        scales = range(4) if self.edges_depth_edge_loss_all_scales else range(1)
        for s in scales:
            sfx = '' if s == 0 else '_%d' % s
            loss, _ = self.edge_loss(inv_depths[s], batch['edge' + sfx], gt_mask=seg_mask, is_grad=is_grad, is_sigmoid=is_sigmoid,
                                     sigmoid_thresh=sigmoid_thresh, gt_normals=batch.get('normal' + sfx),
                                     from_inv_depth=True, return_grad_map=False)
            total = loss if total is None else total + loss
        return total / 4 if self.edges_depth_edge_loss_all_scales else total

    def _fused_losses(self, inv_depths, batch):
        """Edge loss of all scales + silog in one forward / one backward launch (kernels.DepthLossesFn): the shipped
        configuration (4 scales, cross-entropy head, sparse-silog on scale 0).  -> (edge_loss, supervised 'loss' [1]) or None when
        the configuration needs the per-scale path."""
        from .. import kernels as K
        from ..losses.grad_loss import GradLoss
        head = getattr(self, 'edge_loss_head', None)
        sup = self._supervised_loss
        # the fused launch IS GradLoss('cross_entropy') on four scales + 'sparse-silog' on scale 0: anything else (another head
        # class, another edge loss type, another supervised method / scale count) takes the per-scale path and its own errors
        if (not self.edges_depth_edge_loss_all_scales or not isinstance(head, GradLoss) or head.edge_loss_type != 'cross_entropy'
                or getattr(sup, 'supervised_method', None) != 'sparse-silog' or getattr(sup, 'n', 0) != 1
                or not inv_depths[0].is_cuda or tuple(batch['depth'].shape[-2:]) != tuple(inv_depths[0].shape[-2:])):
            return None
        sfx = ['', '_1', '_2', '_3']
        edges = [batch['edge' + s] for s in sfx]
        normals = [batch.get('normal' + s) for s in sfx]
        if any(tuple(e.shape[-2:]) != tuple(i.shape[-2:]) for e, i in zip(edges, inv_depths)):
            return None
        losses = K.DepthLossesFn.apply(head.weight, head.depth_edges_loss_pos_to_neg_weight, float(self.SIGMOID_THRESH), True, batch.get('rgb_edge'),
                                       batch['depth'], edges, normals, *inv_depths[:4])
        sup.add_metric('supervised_loss', losses[4])
        return losses[:4].sum() / 4, losses[4:5]

    def forward(self, batch, return_logs=False, progress=0.0, **kwargs):
        if not self.training:
            return SfmModel.forward(self, batch, return_logs=return_logs, **kwargs)
        out = SfmModel.forward(self, batch, return_logs=return_logs, **kwargs)
        inv_depths = out['inv_depths']
        fused = self._fused_losses(inv_depths, batch) if self.fuse_losses else None
        if fused is not None:
            edge_loss, sup_loss = fused
        else:
            edge_loss = self.compute_edge_loss_with_all_scales(inv_depths, batch, batch.get('rgb_edge'), is_grad=True,
                                                               is_sigmoid=True, sigmoid_thresh=self.SIGMOID_THRESH)
            sup_loss = self.supervised_loss(inv_depths, batch['depth'], return_logs=return_logs, progress=progress)['loss']
        supervised_loss = self.supervised_loss_weight * sup_loss
        edge_loss = self.depth_edges_loss_weight * edge_loss
        loss = supervised_loss + edge_loss
        metrics = {'metrics': {'edge_loss': edge_loss.detach(), 'supervised_loss': supervised_loss.detach()}}
        return {'loss': loss, **merge_outputs(out, metrics)}
