"""Class registry of the reference (packnet_sfm/utils/load.py:36-114): classes are looked up by NAME in a module of the
same name.  Here the search path is this package, so ``load_class('PackNetSAN01', 'networks.depth')`` returns the
gfx950 implementation; unknown names raise ValueError('Unknown class ...') like the reference."""
import importlib
import importlib.util
from inspect import signature

PACKAGE = 'mindtheedge_amd'
_ALIASES = ('packnet_code.packnet_sfm.', 'packnet_sfm.', PACKAGE + '.')


def make_list(var):
    return var if isinstance(var, (list, tuple)) else [var]


def filter_args(func, keys):
    sign = list(signature(func).parameters.keys())
    return {k: v for k, v in dict(keys).items() if k in sign}


def filter_args_create(func, keys):
    return func(**filter_args(func, keys))


def load_class(filename, paths, concat=True):
    for path in make_list(paths):
        for a in _ALIASES:                       # accept the reference's 'packnet_code.packnet_sfm.models' spelling
            if path.startswith(a):
                path = path[len(a):]
        full = '{}.{}.{}'.format(PACKAGE, path, filename) if concat else '{}.{}'.format(PACKAGE, path)
        try:
            found = importlib.util.find_spec(full)
        except ModuleNotFoundError:
            found = None
        if found:
            return getattr(importlib.import_module(full), filename)
    raise ValueError('Unknown class {}'.format(filename))


def load_class_args_create(filename, paths, args={}, concat=True):
    return filter_args_create(load_class(filename, paths, concat), args)


def load_network(network, path, prefixes=''):
    """Prefix-stripped, shape-checked, non-strict checkpoint load (reference load.py:117-166)."""
    import torch
    ckpt = torch.load(path, map_location='cpu')
    sd = ckpt.get('state_dict', ckpt)
    own = network.state_dict()
    picked = {}
    for key, val in sd.items():
        for prefix in make_list(prefixes):
            for p in ([prefix + '.', 'model.' + prefix + '.'] if prefix else ['']):
                if key.startswith(p):
                    k = key[len(p):]
                    if k in own and tuple(own[k].shape) == tuple(val.shape):
                        picked[k] = val
    network.load_state_dict(picked, strict=False)
    return network
