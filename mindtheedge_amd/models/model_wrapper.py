"""Registry + step API the reference trainer talks to (packnet_sfm/models/model_wrapper.py): ``setup_model``,
``setup_depth_net``, ``setup_depth_edge_loss`` (:561-672), ``configure_optimizers`` (:142-180), ``training_step``
(:197-213), ``depth`` (:318-321), and the depth half of validation: ``evaluate_depth`` (:328-352), ``validation_step``
(:215-225), ``validation_epoch_end`` (:255-277) with the metrics computed on the device.
Dataloaders, metric printing and the Canny/chamfer edge metrics (OpenCV) are out of scope."""
import random
from collections import OrderedDict

import torch
import torch.nn as nn

from ..utils.load import load_class, load_class_args_create, load_network, filter_args
from ..losses.grad_loss import GradLoss
from ..utils.depth import inv2depth, post_process_inv_depth, compute_depth_metrics
from .model_utils import stack_batch, flip_lr


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
        # the depth-edge estimator runs the RGB+LiDAR pass: its network owns the sparse branch (with_san is not a reference key)
        extra = {'with_san': True} if config.model.name.startswith('EdgeEstimation') and 'with_san' not in config.model.depth_net else {}
        model.add_depth_net(setup_depth_net(config.model.depth_net, prepared, **extra))
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
        self.metrics_name = 'depth'
        self.metrics_keys = ('abs_rel', 'sqr_rel', 'rmse', 'rmse_log', 'a1', 'a2', 'a3')
        self.metrics_modes = ('', '_pp', '_gt', '_pp_gt')
        self.model = setup_model(config, prepared=resume is not None)
        self._train_dataloader = self._val_dataloaders = None
        self.trainer = None
        if resume and 'state_dict' in resume:
            # reference :88-94: prefix-stripped, shape-checked, NON-strict (a checkpoint written by the reference always
            # carries the sparse-branch tensors model.depth_net.mconvs.*, which this network only owns when with_san=True);
            # 'epoch' is the 0-based index of the epoch that had finished when the file was written -> resume at epoch + 1
            self.model = load_network(self.model, resume['state_dict'], 'model')
            if 'epoch' in resume:
                self.current_epoch = resume['epoch'] + 1

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
        from ..trainers.data_parallel import (FlatParameters, BucketedAllReduce, FusedAdam, broadcast_parameters,
                                              reference_parameter_names)
        opt_cfg = self.config.model.optimizer
        if opt_cfg.name != 'Adam' or opt_cfg.depth.get('weight_decay', 0.0) not in (0, 0.0):
            raise NotImplementedError("the shipped configs use Adam without weight decay")
        flat = FlatParameters(self.depth_net.parameters())
        reducer = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            broadcast_parameters(flat, group=process_group)
            reducer = BucketedAllReduce(flat, process_group)
        optimizer = FusedAdam(flat, lr=opt_cfg.depth.lr, reducer=reducer, name='Depth')
        optimizer.set_index_space(reference_parameter_names(self.depth_net), dict(self.depth_net.named_parameters()))
        sched = getattr(torch.optim.lr_scheduler, self.config.model.scheduler.name)
        scheduler = sched(optimizer, **filter_args(sched, self.config.model.scheduler))
        if self.resume:
            if 'optimizer' in self.resume:
                optimizer.load_state_dict(self.resume['optimizer'])
            if 'scheduler' in self.resume:
                scheduler.load_state_dict(self.resume['scheduler'])
        self.optimizer, self.scheduler = optimizer, scheduler
        return optimizer, scheduler

    # dataset readers are outside this build's scope: the entry point hands the loaders over and the
    # trainer asks for them the way the reference's does (trainers/common_trainer.py:66-67)
    def set_dataloaders(self, train=None, val=None):
        self._train_dataloader, self._val_dataloaders = train, val

    def train_dataloader(self):
        if self._train_dataloader is None:
            raise RuntimeError("no training data: call set_dataloaders(train=...) (train_edges.py --synthetic / --data)")
        return self._train_dataloader

    def val_dataloader(self):
        return self._val_dataloaders or []

    def training_step(self, batch, *args):
        batch = stack_batch(batch)
        output = self.model(batch, progress=self.progress)
        return {'loss': output['loss'], 'metrics': output['metrics']}

    @torch.no_grad()
    def evaluate_depth(self, batch, args=None):
        """Depth metrics of one validation batch, entirely on the device: prediction, prediction on the mirrored
        input, flip-TTA fusion, and the 7 metrics x 4 modes ('', '_pp', '_gt', '_pp_gt') -- reference :328-352.
        The depth metrics stay DEVICE tensors (float32[7]) and never synchronise with the host; the optional edge metrics
        (batch carries 'edge') read one convergence flag per 8 hysteresis sweeps of the Canny step."""
        inv_depths = self.model(batch)['inv_depths'][0]
        inv_depth = inv_depths[0][:, 0:1, :, :]
        depth = inv2depth(inv_depth)
        flipped = dict(batch)
        for key in ('rgb', 'input_depth', 'rgb_edge'):
            if key in flipped:
                flipped[key] = flip_lr(flipped[key])
        inv_depth_flipped = self.model(flipped)['inv_depths'][0][0][:, 0:1, :, :]
        inv_depth_pp = post_process_inv_depth(inv_depth, inv_depth_flipped, method='mean')
        depth_pp = inv2depth(inv_depth_pp)
        metrics = OrderedDict()
        if 'depth' in batch:
            for mode in self.metrics_modes:
                metrics[self.metrics_name + mode] = compute_depth_metrics(
                    self.config.model.params, gt=batch['depth'], pred=depth_pp if 'pp' in mode else depth,
                    use_gt_scale='gt' in mode)
        if 'edge' in batch:
            # reference :354-371 + compute_edge_metrics :373-440: Canny on the FIRST image's predicted depth (three settings)
            # (resized to the edge image's size) against its ground-truth edge image -> (precision, recall, F1) x 3.  The
            # resize and Canny steps restate OpenCV's published algorithms and are parity-unpinned (oracle/canny_oracle.py);
            # the chamfer part is pinned.
            from ..utils.edge import compute_edge_metrics_from_depth
            gt_edge = batch['edge'][0, 0].float() * 255
            metrics['edges'] = compute_edge_metrics_from_depth(depth[0, 0], gt_edge)
        return {'metrics': metrics, 'inv_depth': inv_depth_pp}

    def validation_step(self, batch, *args):
        output = self.evaluate_depth(stack_batch(batch), args)
        return {'idx': batch.get('idx'), **output['metrics']}

    def validation_epoch_end(self, output_data_batch):
        """Mean of the per-batch metrics over the epoch (and over ranks) -> {'depth-abs_rel_pp_gt': float, ...}.
        One host read per epoch."""
        import torch.distributed as dist
        names = [self.metrics_name + m for m in self.metrics_modes]
        rows = [torch.stack([o[n] for n in names]) for o in output_data_batch if all(n in o for n in names)]
        dev = rows[0].device if rows else (torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else 'cpu')
        total = torch.stack(rows).sum(0) if rows else torch.zeros((len(names), len(self.metrics_keys)), device=dev)
        count = torch.tensor(float(len(rows)), device=dev)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(total)                           # every rank takes part, also one whose shard was empty
            dist.all_reduce(count)
        if float(count) == 0.0:
            return {}
        mean = (total / count).cpu()
        from .. import kernels as K
        K.check_device_errors()                              # the validation forwards of the epoch have finished (host read above)
        out = {'{}-{}{}'.format(self.metrics_name, key, mode): float(mean[i, j])
               for i, mode in enumerate(self.metrics_modes) for j, key in enumerate(self.metrics_keys)}
        edge_rows = [o['edges'] for o in output_data_batch if 'edges' in o]
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if edge_rows or multi:                               # (precision, recall, F1) x the three Canny settings, reference :431-438
            etotal = torch.stack(edge_rows).double().sum(0) if edge_rows else torch.zeros(9, dtype=torch.float64, device=dev)
            ecount = torch.tensor(float(len(edge_rows)), device=etotal.device, dtype=etotal.dtype)
            if multi:                                        # unconditional: a rank without edge rows must not leave the others waiting
                dist.all_reduce(etotal)
                dist.all_reduce(ecount)
        if (edge_rows or multi) and float(ecount) > 0:
            emean = (etotal / ecount).cpu()
            for k in range(emean.numel() // 3):
                for j, key in enumerate(('precision', 'recall', 'f1')):
                    out['edges-{}_{}'.format(key, k)] = float(emean[3 * k + j])
        return out

    def depth(self, *args, **kwargs):
        out = self.model.depth_net(*args, **kwargs)
        if not self.model.depth_net.training:
            # inference has no optimizer step to poll the device error word (a GroupNorm cluster wait that gave up): one read of host memory, no
            # synchronisation -- a give-up of THIS forward surfaces at the latest at the next call (infer_edges.infer_depth waits and polls itself)
            from .. import kernels as K
            K.check_device_errors()
        return out

    def forward(self, *args, **kwargs):
        return self.model(*args, **kwargs)
