#!/usr/bin/env python3
"""train_edges.py <config.yaml> -- entry point with the reference's calling convention (/root/reference/train_edges.py:17-69):
parse the YAML over the defaults, build ModelWrapper (registry -> SemiSupEdgeModel + PackNetSAN01 + GradLoss), fit.
Multi-GPU: `python -m torch.distributed.run --nproc-per-node N train_edges.py <yaml>` (one process per GPU, RCCL).
Datasets are outside this build's scope: `--synthetic` trains on synthetic frames of the
configured image_shape; otherwise pass `--data module:callable` returning an iterable of batch dicts."""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # see mindtheedge_amd/__init__.py: side stream / RCCL streams need their own hardware queues


def main():
    ap = argparse.ArgumentParser(description='PackNet-SAN + depth-edge-loss training on MI355X')
    ap.add_argument('file', type=str, help='Input file (.yaml)')
    ap.add_argument('--synthetic', action='store_true')
    ap.add_argument('--steps', type=int, default=20, help='steps per epoch for --synthetic')
    ap.add_argument('--synthetic-lidar', action='store_true', help="add a sparse 'input_depth' to the synthetic batches (DEE training with LiDAR)")
    ap.add_argument('--epochs', type=int, default=1)
    ap.add_argument('--data', type=str, default=None)
    ap.add_argument('--resume', type=str, default=None, help='.ckpt in the reference layout (model_checkpoint.py:71-81) to resume from')
    ap.add_argument('--save', type=str, default=None, help='write <save>/epoch=N.ckpt after every epoch (rank 0)')
    args = ap.parse_args()
    assert args.file.endswith('.yaml'), 'You need to provide a .yaml file'
    import torch
    from mindtheedge_amd.utils.config import load_config
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    from mindtheedge_amd.trainers.trainer import Trainer
    from mindtheedge_amd.utils.synthetic import SyntheticLoader
    config = load_config(args.file)
    if config.model.depth_net.checkpoint_path and not os.path.exists(config.model.depth_net.checkpoint_path):
        print('WARNING: depth_net.checkpoint_path %r does not exist -- training from the xavier initialisation'
              % config.model.depth_net.checkpoint_path, file=sys.stderr)
        config.model.depth_net.checkpoint_path = ''
    from mindtheedge_amd.models.model_checkpoint import load_checkpoint
    save_dir = args.save or (os.path.dirname(config.checkpoint.filepath) if config.checkpoint.filepath else None)
    trainer = Trainer(**{**config.arch, 'checkpoint': save_dir})
    wrapper = ModelWrapper(config, resume=load_checkpoint(args.resume) if args.resume else None)
    import ast
    H, W = tuple(config.datasets.augmentation.image_shape) if not isinstance(config.datasets.augmentation.image_shape, str) \
        else ast.literal_eval(config.datasets.augmentation.image_shape)
    if args.data:
        mod, fn = args.data.split(':')
        loader = getattr(importlib.import_module(mod), fn)(config, trainer.proc_rank, trainer.world_size)
    else:
        assert args.synthetic, 'no dataset: pass --synthetic or --data module:callable'
        loader = SyntheticLoader(config.datasets.train.batch_size, H, W, args.steps, trainer.device, trainer.proc_rank,
                                 lidar=args.synthetic_lidar)
    wrapper.set_dataloaders(train=loader)
    trainer.max_epochs = wrapper.current_epoch + args.epochs
    hist = trainer.fit(wrapper)
    if trainer.is_rank_0:
        print(hist)


if __name__ == '__main__':
    main()
