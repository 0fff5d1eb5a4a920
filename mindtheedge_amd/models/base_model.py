"""The small bookkeeping surface every packnet_sfm model exposes to ModelWrapper / the trainer (reference:
packnet_sfm/models/base_model.py:7-97): which networks a model needs (``network_requirements``), which batch keys
training needs (``train_requirements``), the batch keys forwarded to the depth network (``_input_keys``), and read-only
views of the detached ``logs`` / ``losses`` dictionaries."""
import torch.nn as nn


def _read_only(attr):
    return property(lambda self: getattr(self, attr))


class BaseModel(nn.Module):
    logs = _read_only('_logs')
    losses = _read_only('_losses')
    network_requirements = _read_only('_network_requirements')
    train_requirements = _read_only('_train_requirements')

    def __init__(self, **kwargs):
        super().__init__()
        self._logs, self._losses = {}, {}
        self._network_requirements, self._train_requirements = [], []
        self._input_keys = ['rgb']

    def add_loss(self, key, val):
        self._losses[key] = val.detach()

    def add_net(self, network_module, network_name):
        if network_name not in self._network_requirements:
            raise AssertionError("Network module not required!")
        setattr(self, network_name, network_module)

    def forward(self, batch, return_logs=False, **kwargs):
        raise NotImplementedError("subclasses implement forward(batch)")
