"""BaseModel API of the packnet_sfm model wrapper (reference: packnet_sfm/models/base_model.py:7-97)."""
import torch.nn as nn


class BaseModel(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self._logs = {}
        self._losses = {}
        self._network_requirements = []
        self._train_requirements = []
        self._input_keys = ['rgb']

    @property
    def logs(self):
        return self._logs

    @property
    def losses(self):
        return self._losses

    def add_loss(self, key, val):
        self._losses[key] = val.detach()

    @property
    def network_requirements(self):
        return self._network_requirements

    @property
    def train_requirements(self):
        return self._train_requirements

    def add_net(self, network_module, network_name):
        assert network_name in self._network_requirements, "Network module not required!"
        setattr(self, network_name, network_module)

    def forward(self, batch, return_logs=False, **kwargs):
        raise NotImplementedError("Please implement forward function in your own subclass model.")
