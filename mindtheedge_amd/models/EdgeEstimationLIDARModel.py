"""EdgeEstimationLIDARModel (the DEE: depth-edge estimator with an optional LiDAR input), INFERENCE side -- drop-in for
packnet_sfm/models/EdgeEstimationLIDARModel.py:28-181 as used by the annotation config
(configs/annotate_edges_kitti_training_set.yaml: ``model.name: 'EdgeEstimationLIDARModel'``).

Eval-mode forward, as the reference (:104-133): ``input_depth / 200`` -> SfmModel.forward (the network takes the
RGB+LiDAR pass through the sparse SAN branch when ``input_depth`` is present) -> the full-resolution output ``/ 2`` is
the edge probability.  Training the DEE needs a backward pass through the SAN branch (batch statistics over the active
points, pooling and fusion gradients) plus the RGB/RGB-D feature-consistency loss (:139-160); that is not built and
``forward`` raises in training mode instead of computing something else.  The SAN branch is parity-unpinned
(networks/layers/minkowski_encoder.py).
"""
from .SfmModel import SfmModel


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

    def forward(self, batch, return_logs=False, progress=0.0, **kwargs):
        if self.training:
            raise NotImplementedError("training the DEE needs the backward pass of the sparse SAN branch (SURVEY.md 8 f-1/f-2), "
                                      "which this build does not have; use eval mode for annotation / inference")
        batch = dict(batch)
        if 'input_depth' in batch:
            batch['input_depth'] = batch['input_depth'] / 200.0          # reference :108-110 ("why 200?")
        out = SfmModel.forward(self, batch, return_logs=return_logs, **kwargs)
        inv = out['inv_depths']
        scales = inv[0] if isinstance(inv[0], list) else inv             # eval: [[scale0..3], features]
        scales[0] = scales[0] / 2                                        # reference :119-124 (num_scales = 1 in eval)
        return out
