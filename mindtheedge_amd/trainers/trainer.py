"""One-process-per-GPU trainer: the MI355X replacement of CommonTrainer/HorovodTrainer
(packnet_sfm/trainers/common_trainer.py:22-185, horovod_trainer.py).  Same surface: ``Trainer(**config.arch,
checkpoint=None).fit(module)``, ``proc_rank``, ``world_size``, ``is_rank_0``.  The inner loop is the reference's
zero_grad -> to-device -> training_step -> backward -> optimizer.step, minus its 4-6 ``.item()`` host syncs per step:
running means are accumulated on the device and read once per ``log_every`` steps."""
import os

import torch
import torch.distributed as dist


def sample_to_cuda(data, device):
    if isinstance(data, dict):
        return {k: sample_to_cuda(v, device) for k, v in data.items()}
    if isinstance(data, (list, tuple)):
        return [sample_to_cuda(v, device) for v in data]
    if torch.is_tensor(data):
        return data.to(device, non_blocking=True)
    return data


class Trainer:
    def __init__(self, min_epochs=0, max_epochs=50, validate_first=False, checkpoint=None, log_every=50, **kwargs):
        self.min_epochs, self.max_epochs, self.validate_first = min_epochs, max_epochs, validate_first
        self.checkpoint, self.log_every = checkpoint, log_every
        self.distributed = int(os.environ.get('WORLD_SIZE', '1')) > 1
        if self.distributed and not dist.is_initialized():
            local = int(os.environ.get('LOCAL_RANK', '0'))
            torch.cuda.set_device(local)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        self.device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None

    @property
    def proc_rank(self):
        return dist.get_rank() if dist.is_initialized() else 0

    @property
    def world_size(self):
        return dist.get_world_size() if dist.is_initialized() else 1

    @property
    def is_rank_0(self):
        return self.proc_rank == 0

    def fit(self, module, train_dataloader=None, epochs=None, val_dataloaders=None):
        """``fit(module)`` as the reference (trainers/common_trainer.py:42-91): the module hands over its optimizer, scheduler
        and dataloaders (``module.train_dataloader()`` / ``module.val_dataloader()``); the loaders may also be passed in."""
        if self.device is None:
            raise RuntimeError("mindtheedge_amd trains on MI355X GPUs only")
        module.trainer = self
        module.to(self.device)
        optimizer, scheduler = module.configure_optimizers()
        if train_dataloader is None:
            train_dataloader = module.train_dataloader()
        if val_dataloaders is None:
            val_dataloaders = module.val_dataloader() if hasattr(module, 'val_dataloader') else None
        history = []
        if val_dataloaders and self.validate_first:
            history.append({'validation': self.validate(val_dataloaders, module)})
        for epoch in range(module.current_epoch, epochs if epochs is not None else self.max_epochs):
            if hasattr(getattr(train_dataloader, 'sampler', None), 'set_epoch'):
                train_dataloader.sampler.set_epoch(epoch)
            history.append(self.train(train_dataloader, module, optimizer))
            if val_dataloaders:
                history[-1]['validation'] = self.validate(val_dataloaders, module)
            # as the reference (:80-91): save BEFORE the epoch counter advances, so 'epoch' in the file is the 0-based index of
            # the epoch that just finished and a resume continues at epoch + 1 (one file per epoch, no top-k policy)
            if self.checkpoint and self.is_rank_0:
                from ..models.model_checkpoint import save_checkpoint
                save_checkpoint(os.path.join(str(self.checkpoint), 'epoch={}.ckpt'.format(module.current_epoch)), module)
            module.current_epoch += 1
            scheduler.step()
        return history

    def validate(self, dataloaders, module):
        """Reference CommonTrainer.validate (trainers/common_trainer.py:187-210): every validation dataset in turn,
        ``module.validation_step`` per batch, ``module.validation_epoch_end`` per dataset -> [metrics dict per dataset].
        The per-batch metrics stay on the device; the epoch end reads them once."""
        was_training = module.training
        module.eval()
        results = []
        for n, dataloader in enumerate(dataloaders):
            outputs = [module.validation_step(sample_to_cuda(batch, self.device), i, n) for i, batch in enumerate(dataloader)]
            results.append(module.validation_epoch_end(outputs))
        module.train(was_training)
        return results

    def train(self, dataloader, module, optimizer):
        module.train()
        run = torch.zeros(3, device=self.device)               # loss, supervised, edge running sums (device side)
        n, last = 0, None
        for i, batch in enumerate(dataloader):
            optimizer.zero_grad()
            batch = sample_to_cuda(batch, self.device)
            output = module.training_step(batch, i, None)
            output['loss'].backward()
            optimizer.step()
            m = output['metrics']
            zero = run.new_zeros(())
            run += torch.stack([output['loss'].detach().sum(), m.get('supervised_loss', zero).sum(), m.get('edge_loss', zero).sum()])
            n += 1
            if self.is_rank_0 and self.log_every and n % self.log_every == 0:
                last = (run / n).tolist()                       # the only host sync
                print('step {} | Avg. {:.4f} | Sup. {:.4f} | Edge RGB {:.4f}'.format(n, *last), flush=True)
        if n:
            last = (run / n).tolist()
        return {'steps': n, 'avg_loss': last[0] if last else None, 'avg_supervised': last[1] if last else None,
                'avg_edge': last[2] if last else None}
