"""Registry + step API the reference trainer talks to (packnet_sfm/models/model_wrapper.py): ``setup_model``,
``setup_depth_net``, ``setup_depth_edge_loss`` (:561-672), ``configure_optimizers`` (:142-180), ``training_step``
(:197-213), ``depth`` (:318-321).  Dataloaders / metric printing are out of scope."""
import random

import torch
import torch.nn as nn

from ..utils.load import load_class, load_class_args_create, load_network, filter_args
from ..losses.grad_loss import GradLoss
from .model_utils import stack_batch


def set_random_seed(seed):
    if seed >= 0:
        random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)


def setup_depth_net(config, prepared, **kwargs):
    depth_net = load_class_args_create(config.name, paths=['networks.depth'], args={**config, **kwargs})
    if not prepared and config.checkpoint_path != '':
        depth_net = load_network(depth_net, config.checkpoint_path, ['depth_net', 'disp_network'])
    return depth_net


def setup_depth_edge_loss(config):
    return GradLoss(config.edges.edge_loss_type, config.edges.use_external_edges_for_loss,
                    config.edges.edge_loss_class_list_to_mask_out, config.edges.depth_edges_loss_weight,
                    config.edges.depth_edge_loss_pos_to_neg_weight)


def setup_model(config, prepared, **kwargs):
    model = load_class(config.model.name, paths=['models'])(**{**config.model.loss, **kwargs})
    if 'depth_net' in model.network_requirements:
        model.add_depth_net(setup_depth_net(config.model.depth_net, prepared))
    if 'pose_net' in model.network_requirements:
        raise NotImplementedError("pose networks are outside this build's scope")
    if config.edges.train_depth_edges:
        model.add_edge_loss(setup_depth_edge_loss(config))
    if not prepared and config.model.checkpoint_path != '':
        model = load_network(model, config.model.checkpoint_path, 'model')
    if config.is_multi_gpu:
        raise NotImplementedError("nn.DataParallel is replaced by one process per GPU: launch with torch.distributed.run")
    return model


class ModelWrapper(nn.Module):
    def __init__(self, config, resume=None, logger=None, load_datasets=False):
        super().__init__()
        self.config, self.logger, self.resume = config, logger, resume
        set_random_seed(config.arch.seed)
        self.model = self.optimizer = self.scheduler = None
        self.current_epoch = 0
        self.model = setup_model(config, prepared=resume is not None)
        if resume and 'state_dict' in resume:
            self.load_state_dict(resume['state_dict'])
            self.current_epoch = resume.get('epoch', 0)

    @property
    def depth_net(self):
        return self.model.depth_net

    @property
    def pose_net(self):
        return getattr(self.model, 'pose_net', None)

    @property
    def progress(self):
        return self.current_epoch / self.config.arch.max_epochs

    def configure_optimizers(self, process_group=None):
        """Adam over depth_net.parameters() (param group 'Depth') + StepLR, as the reference; the optimizer is the
        fused flat Adam and carries the bucketed RCCL gradient averaging when torch.distributed is initialised."""
        import torch.distributed as dist
        from ..trainers.data_parallel import FlatParameters, BucketedAllReduce, FusedAdam, broadcast_parameters
        opt_cfg = self.config.model.optimizer
        if opt_cfg.name != 'Adam' or opt_cfg.depth.get('weight_decay', 0.0) not in (0, 0.0):
            raise NotImplementedError("the shipped configs use Adam without weight decay")
        flat = FlatParameters(self.depth_net.parameters())
        reducer = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            broadcast_parameters(flat, group=process_group)
            reducer = BucketedAllReduce(flat, process_group)
        optimizer = FusedAdam(flat, lr=opt_cfg.depth.lr, reducer=reducer, name='Depth')
        sched = getattr(torch.optim.lr_scheduler, self.config.model.scheduler.name)
        scheduler = sched(optimizer, **filter_args(sched, self.config.model.scheduler))
        if self.resume:
            if 'optimizer' in self.resume:
                optimizer.load_state_dict(self.resume['optimizer'])
            if 'scheduler' in self.resume:
                scheduler.load_state_dict(self.resume['scheduler'])
        self.optimizer, self.scheduler = optimizer, scheduler
        return optimizer, scheduler

    def training_step(self, batch, *args):
        batch = stack_batch(batch)
        output = self.model(batch, progress=self.progress)
        return {'loss': output['loss'], 'metrics': output['metrics']}

    def depth(self, rgb, **kwargs):
        return self.model.depth_net(rgb=rgb, **kwargs)

    def forward(self, *args, **kwargs):
        return self.model(*args, **kwargs)
