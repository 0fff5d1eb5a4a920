"""EdgeEstimationLIDARModel (the DEE: depth-edge estimator with an optional LiDAR input) -- drop-in for
packnet_sfm/models/EdgeEstimationLIDARModel.py:28-181 as used by the annotation config
(configs/annotate_edges_kitti_training_set.yaml: ``model.name: 'EdgeEstimationLIDARModel'``).

Eval-mode forward, as the reference (:104-133): ``input_depth / 200`` -> SfmModel.forward (the network takes the
RGB+LiDAR pass through the sparse SAN branch when ``input_depth`` is present) -> the full-resolution output ``/ 2`` is
the edge probability.

Training (reference :135-160): loss = depth_loss + (edge_rgb + weight_rgbd * edge_lidar) / 2, each edge term the mean over
the 4 scales of the balanced BCE of ``inv_depth_s / 2`` against ``edge_s`` taken directly on the probability (is_grad=False,
is_sigmoid=False), on the same HIP kernels as the depth path.  Without ``input_depth`` in the batch only the RGB term exists;
with it the network runs its second, RGB+LiDAR pass through the sparse SAN branch (PackNetSAN01(with_san=True)), whose
backward pass and the feature-matching ``depth_loss`` are built in round 2.  The SAN branch is parity-unpinned
(networks/layers/minkowski_encoder.py).
"""
from .SfmModel import SfmModel
from .model_utils import merge_outputs


class EdgeEstimationLIDARModel(SfmModel):
    _train_with_lidar = True

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
        batch = dict(batch)
        if 'input_depth' in batch:
            batch['input_depth'] = batch['input_depth'] / 200.0          # reference :108-110 ("why 200?")
        if self.training:
            out = SfmModel.forward(self, batch, return_logs=return_logs, **kwargs)
            n = 4 if self.edges_depth_edge_loss_all_scales else 1

            def halve(invs):                                             # reference :119-131
                return [inv / 2 for inv in invs[:n]] + list(invs[n:])

            probs = halve(out['inv_depths'])
            out = {**out, 'inv_depths': probs}
            edge_rgb_loss = self.compute_edge_loss_with_all_scales(probs, batch, None, is_grad=False, is_sigmoid=False)
            metrics = {'edge_loss': edge_rgb_loss.detach()}
            loss = 0.0
            edge_lidar_loss = 0.0
            if 'inv_depths_rgbd' in out:                                 # two-pass RGB / RGB+LiDAR step (reference :140-153)
                probs_rgbd = halve(out['inv_depths_rgbd'])
                out['inv_depths_rgbd'] = probs_rgbd
                edge_lidar_loss = self.compute_edge_loss_with_all_scales(probs_rgbd, batch, None, is_grad=False, is_sigmoid=False)
                metrics['edge_lidar_loss'] = edge_lidar_loss.detach()
                if 'depth_loss' in out:
                    loss = loss + out['depth_loss']
            loss = loss + (edge_rgb_loss + self.weight_rgbd * edge_lidar_loss) / 2
            return {'loss': loss, **merge_outputs(out, {'metrics': metrics})}
        out = SfmModel.forward(self, batch, return_logs=return_logs, **kwargs)
        inv = out['inv_depths']
        scales = inv[0] if isinstance(inv[0], list) else inv             # eval: [[scale0..3], features]
        scales[0] = scales[0] / 2                                        # reference :119-124 (num_scales = 1 in eval)
        return out
