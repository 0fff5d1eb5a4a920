"""EdgeEstimationLIDARModel (the DEE: depth-edge estimator with an optional LiDAR input), INFERENCE side -- drop-in for
packnet_sfm/models/EdgeEstimationLIDARModel.py:28-181 as used by the annotation config
(configs/annotate_edges_kitti_training_set.yaml: ``model.name: 'EdgeEstimationLIDARModel'``).

Eval-mode forward, as the reference (:104-133): ``input_depth / 200`` -> SfmModel.forward (the network takes the
RGB+LiDAR pass through the sparse SAN branch when ``input_depth`` is present) -> the full-resolution output ``/ 2`` is
the edge probability.

Training: the RGB-only form (no ``input_depth`` in the batch; reference :135-160 with ``edge_lidar_loss = 0``) is built --
loss = (mean over the 4 scales of the balanced BCE of ``inv_depth_s / 2`` against ``edge_s``, taken directly on the
probability: is_grad=False, is_sigmoid=False) / 2 -- on the same HIP kernels as the depth path.  Training WITH a LiDAR
input needs a backward pass through the SAN branch (batch statistics over the active points, pooling and fusion
gradients) plus the RGB/RGB-D feature-consistency loss; that is not built and ``forward`` raises instead of computing
something else.  The SAN branch is parity-unpinned (networks/layers/minkowski_encoder.py).
"""
from .SfmModel import SfmModel
from .model_utils import merge_outputs


class EdgeEstimationLIDARModel(SfmModel):
    def __init__(self, supervised_loss_weight=0.0, weight_rgbd=1.0, **kwargs):
        super().__init__(**kwargs)
        self.supervised_loss_weight = supervised_loss_weight
        self.weight_rgbd = weight_rgbd
        self._network_requirements.remove('pose_net')
        self._train_requirements.append('gt_depth')
        self.edges_depth_edge_loss_all_scales = kwargs.get('edges_depth_edge_loss_all_scales', False)
        self._input_keys = ['rgb', 'input_depth', 'edge']
        if self.edges_depth_edge_loss_all_scales:
            self._input_keys += ['edge_1', 'edge_2', 'edge_3']

    def edge_loss(self, pred, gt_edges, gt_mask=None, is_grad=True, is_sigmoid=False, sigmoid_thresh=0):
        return self.edge_loss_head(pred, gt_edges, gt_mask, is_grad, is_sigmoid, sigmoid_thresh)

    def compute_edge_loss_with_all_scales(self, probs, batch, seg_mask, is_grad=False, is_sigmoid=False):
        """reference :163-181: BCE of the prediction at every scale against edge / edge_1..3, averaged over the 4 scales"""
        total, _ = self.edge_loss(probs[0], batch['edge'], gt_mask=seg_mask, is_grad=is_grad, is_sigmoid=is_sigmoid)
        if self.edges_depth_edge_loss_all_scales:
            for s in range(1, 4):
                cur, _ = self.edge_loss(probs[s], batch['edge_%d' % s], gt_mask=seg_mask, is_grad=is_grad, is_sigmoid=is_sigmoid)
                total = total + cur
            total = total / 4
        return total

    def forward(self, batch, return_logs=False, progress=0.0, **kwargs):
        if self.training:
            if 'input_depth' in batch:
                raise NotImplementedError("training the DEE with a LiDAR input needs the backward pass of the sparse SAN branch "
                                          "(SURVEY.md 8 f-1/f-2), which this build does not have; drop 'input_depth' to train the "
                                          "RGB-only estimator, or use eval mode for annotation / inference")
            out = SfmModel.forward(self, batch, return_logs=return_logs, **kwargs)
            n = 4 if self.edges_depth_edge_loss_all_scales else 1
            probs = [inv / 2 for inv in out['inv_depths'][:n]] + list(out['inv_depths'][n:])       # reference :119-124
            out = {**out, 'inv_depths': probs}
            edge_rgb_loss = self.compute_edge_loss_with_all_scales(probs, batch, None, is_grad=False, is_sigmoid=False)
            loss = edge_rgb_loss / 2                                                             # (rgb + weight_rgbd * 0) / 2, :153
            return {'loss': loss, **merge_outputs(out, {'metrics': {'edge_loss': edge_rgb_loss.detach()}})}
        batch = dict(batch)
        if 'input_depth' in batch:
            batch['input_depth'] = batch['input_depth'] / 200.0          # reference :108-110 ("why 200?")
        out = SfmModel.forward(self, batch, return_logs=return_logs, **kwargs)
        inv = out['inv_depths']
        scales = inv[0] if isinstance(inv[0], list) else inv             # eval: [[scale0..3], features]
        scales[0] = scales[0] / 2                                        # reference :119-124 (num_scales = 1 in eval)
        return out
