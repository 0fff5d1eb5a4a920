"""SemiSupEdgeModel: silog supervision + multi-scale depth-edge loss over the PackNet-SAN output -- drop-in
for packnet_sfm/models/SemiSupEdgeModel.py (forward :98-162, compute_edge_loss_with_all_scales :164-198).

loss = supervised_loss_weight * silog(inv_0, depth) + depth_edges_loss_weight * mean_s GradLoss(inv2depth(inv_s), edge_s, normal_s)

Differences from the reference, all behaviour-preserving for the shipped configuration:
  * 'input_depth' is not forwarded to the depth network: the RGB+LiDAR pass it triggers upstream never reaches
    the loss (only self_sup_output['inv_depths'] is consumed, reference :130,145).
  * inv2depth is fused into the edge-loss stencil (from_inv_depth=True), and the detached edge-strength maps the
    reference computes and drops are not written.
  * the caller's inv_depths list is not mutated by the supervised loss.
"""
import torch

from .SfmModel import SfmModel
from .model_utils import merge_outputs
from ..losses.supervised_loss import SupervisedLoss


class SemiSupEdgeModel(SfmModel):
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

    def forward(self, batch, return_logs=False, progress=0.0, **kwargs):
        if not self.training:
            return SfmModel.forward(self, batch, return_logs=return_logs, **kwargs)
        out = SfmModel.forward(self, batch, return_logs=return_logs, **kwargs)
        inv_depths = out['inv_depths']
        edge_loss = self.compute_edge_loss_with_all_scales(inv_depths, batch, batch.get('rgb_edge'), is_grad=True,
                                                           is_sigmoid=True, sigmoid_thresh=4)
        sup = self.supervised_loss(inv_depths, batch['depth'], return_logs=return_logs, progress=progress)
        supervised_loss = self.supervised_loss_weight * sup['loss']
        edge_loss = self.depth_edges_loss_weight * edge_loss
        loss = supervised_loss + edge_loss
        metrics = {'metrics': {'edge_loss': edge_loss.detach(), 'supervised_loss': supervised_loss.detach()}}
        return {'loss': loss, **merge_outputs(out, metrics)}
