"""SfmModel: depth network + optional whole-batch horizontal-flip augmentation
(reference: packnet_sfm/models/SfmModel.py:12-133).  The pose network slot exists for API parity only."""
import random

from .base_model import BaseModel
from .model_utils import flip_batch_input, flip_output


class SfmModel(BaseModel):
    def __init__(self, depth_net=None, pose_net=None, rotation_mode='euler', flip_lr_prob=0.0,
                 upsample_depth_maps=False, **kwargs):
        super().__init__()
        self.depth_net = depth_net
        self.pose_net = pose_net
        self.rotation_mode = rotation_mode
        self.flip_lr_prob = flip_lr_prob
        self.upsample_depth_maps = upsample_depth_maps
        if upsample_depth_maps:
            raise NotImplementedError("upsample_depth_maps=True is not used by the shipped edge-loss configs")
        self._network_requirements = ['depth_net', 'pose_net']

    def add_depth_net(self, depth_net):
        self.depth_net = depth_net

    def add_pose_net(self, pose_net):
        self.pose_net = pose_net

    def add_edge_loss(self, edge_loss_head):
        self.edge_loss_head = edge_loss_head

    def depth_net_flipping(self, batch, flip, output_features=False):
        batch_input = {key: batch[key] for key in self._input_keys if key in batch}
        # the reference also forwards 'input_depth' (SemiSupEdgeModel.py:44).  While training, the RGB+LiDAR pass it triggers
        # never reaches the loss and is skipped here; in eval mode it IS the prediction the reference validates, and it is
        # taken when the network owns the sparse branch (PackNetSAN01(with_san=True), parity unpinned)
        batch_input.pop('input_depth', None)
        has_san = getattr(self.depth_net, 'with_san', False)
        if 'input_depth' in batch and ((self.training and self._train_with_lidar) or (not self.training and has_san)):
            batch_input['input_depth'] = batch['input_depth']          # (a network without the branch raises in train mode: no silent RGB-only step)
        batch_input['output_features'] = output_features
        if flip:
            return flip_output(self.depth_net(**flip_batch_input(batch_input)))
        return self.depth_net(**batch_input)

    _train_with_lidar = False    # EdgeEstimationLIDARModel sets it: its loss consumes the RGB+LiDAR pass (SemiSupEdgeModel's does not)

    _pinned_flip = None          # utils.graph.GraphedTrainStep: the host-side flip draw is taken outside the captured step

    def draw_flip(self):
        """the reference's whole-batch flip draw (SfmModel.py:90): python's RNG, one draw per training step"""
        return random.random() < self.flip_lr_prob

    def compute_depth_net(self, batch, force_flip=False, output_features=False):
        if self.training:
            flag_flip_lr = self.draw_flip() if self._pinned_flip is None else self._pinned_flip
        else:
            flag_flip_lr = force_flip
        return self.depth_net_flipping(batch, flag_flip_lr, output_features)

    def forward(self, batch, return_logs=False, force_flip=False, output_features=False):
        depth_output = self.compute_depth_net(batch, force_flip=force_flip, output_features=output_features)
        if 'rgb_context' in batch and self.pose_net is not None:
            raise NotImplementedError("pose networks / self-supervision are outside this build's scope")
        return {**depth_output, 'poses': None}
