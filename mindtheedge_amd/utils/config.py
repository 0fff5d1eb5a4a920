"""Minimal attribute-dict configuration with the reference's defaults for the keys the hot path reads
(packnet_code/configs/default_config.py:8-292; yacs is not required).  ``load_config(yaml_path)`` overlays a reference
YAML (e.g. configs/train_packnet_san_kitti_with_edges.yaml) on the defaults."""
import copy


class Cfg(dict):
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(d):
        if isinstance(d, dict):
            return Cfg({k: Cfg.wrap(v) for k, v in d.items()})
        return d


_DEFAULTS = {
    'is_multi_gpu': False,
    'arch': {'seed': 42, 'min_epochs': 1, 'max_epochs': 50, 'validate_first': False},
    'model': {
        'name': 'SemiSupEdgeModel', 'checkpoint_path': '',
        'optimizer': {'name': 'Adam', 'depth': {'lr': 0.0002, 'weight_decay': 0.0}},
        'scheduler': {'name': 'StepLR', 'step_size': 10, 'gamma': 0.5},
        'params': {'crop': '', 'min_depth': 0.0, 'max_depth': 80.0, 'scale_output': 'resize'},     # default_config.py:84-87
        'loss': {'supervised_method': 'sparse-l1', 'supervised_num_scales': 4, 'supervised_loss_weight': 0.9,
                 'depth_edges_loss_weight': 1.0, 'edges_depth_edge_loss_all_scales': False, 'upsample_depth_maps': False,
                 'flip_lr_prob': 0.5, 'progressive_scaling': 0.0},
        'depth_net': {'name': 'PackNetSAN01', 'checkpoint_path': '', 'version': '1A', 'dropout': 0.0,
                      'freeze_encoder': False, 'freeze_decoder': False, 'freeze_san': False, 'input_channels': 3,
                      'is_depth_aux_net': False, 'output_channels': 1},
    },
    'edges': {'train_depth_edges': True, 'depth_edges_loss_weight': 10.0, 'use_external_edges_for_loss': True,
              'edge_loss_type': 'cross_entropy', 'edge_loss_class_list_to_mask_out': [],
              'depth_edge_loss_pos_to_neg_weight': 1.0},
    'datasets': {'augmentation': {'image_shape': (384, 1280)}, 'train': {'batch_size': 8}},
    'checkpoint': {'filepath': '', 'save_top_k': -1},
}


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def default_config():
    return Cfg.wrap(copy.deepcopy(_DEFAULTS))


def load_config(yaml_path=None, overrides=None):
    cfg = copy.deepcopy(_DEFAULTS)
    if yaml_path:
        import yaml
        with open(yaml_path) as f:
            _merge(cfg, yaml.safe_load(f) or {})
    if overrides:
        _merge(cfg, overrides)
    return Cfg.wrap(cfg)
