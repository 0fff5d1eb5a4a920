"""Checkpoint FORMAT of the reference (checkpoint half): packnet_sfm/models/model_checkpoint.py:71-81
writes ``{'config', 'epoch', 'state_dict', 'optimizer', 'scheduler'}`` with ``state_dict`` keys ``model.depth_net.*`` and
``torch.optim.Adam``'s optimizer layout; packnet_sfm/utils/load.py:117-201 reads it back (and TRI's published
``PackNetSAN01_*.ckpt`` files, whose extra sparse-branch tensors are skipped by the shape-checked, non-strict load in
``utils/load.py::load_network``).  The top-k / S3 / file-naming policy of the reference's ModelCheckpoint is control
plane and is not rebuilt: ``save_checkpoint`` writes one file, ``load_checkpoint`` reads one."""
import os

import torch


def _plain(obj):
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_plain(v) for v in obj)
    return obj


def save_checkpoint(filepath, model_wrapper):
    """One ``.ckpt`` in the reference layout.  Tensors are written from the host (flat-buffer views are cloned)."""
    os.makedirs(os.path.dirname(os.path.abspath(filepath)), exist_ok=True)
    from .. import kernels as K
    if torch.cuda.is_available():
        K.join_side_stream()
        torch.cuda.synchronize()
    ckpt = {
        'config': _plain(dict(model_wrapper.config)),
        'epoch': model_wrapper.current_epoch,
        'state_dict': {k: v.detach().cpu().clone() for k, v in model_wrapper.state_dict().items()},
        'optimizer': model_wrapper.optimizer.state_dict() if model_wrapper.optimizer is not None else None,
        'scheduler': model_wrapper.scheduler.state_dict() if model_wrapper.scheduler is not None else None,
    }
    if ckpt['optimizer'] is not None:
        for st in ckpt['optimizer']['state'].values():
            st['exp_avg'], st['exp_avg_sq'] = st['exp_avg'].cpu(), st['exp_avg_sq'].cpu()
    torch.save(ckpt, filepath)
    return filepath


def load_checkpoint(filepath):
    """-> the checkpoint dict, ready for ``ModelWrapper(config, resume=ckpt)`` (drops None optimizer / scheduler entries)."""
    from ..utils.load import read_checkpoint
    ckpt = read_checkpoint(filepath)
    return {k: v for k, v in ckpt.items() if v is not None}
