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


class _CfgStub(dict):
    """Stands in for yacs.config.CfgNode when a reference checkpoint's 'config' entry is unpickled without yacs."""
    __getattr__ = dict.get


class _TolerantPickle:
    """pickle module for torch.load: classes of packages this image lacks (yacs) become plain dict stand-ins, so the
    tensors of a checkpoint written by the reference can still be read."""
    import pickle as _p
    __name__ = 'pickle'
    load, loads, dump, dumps = _p.load, _p.loads, _p.dump, _p.dumps
    PickleError, UnpicklingError, Pickler = _p.PickleError, _p.UnpicklingError, _p.Pickler

    class Unpickler(_p.Unpickler):
        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                if module.split('.')[0] in ('yacs',):
                    return _CfgStub
                raise


def read_checkpoint(path):
    """torch.load of a `.ckpt` written by this package or by the reference (models/model_checkpoint.py:71-81)."""
    import torch
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except Exception:
        return torch.load(path, map_location='cpu', weights_only=False, pickle_module=_TolerantPickle)


def load_network(network, path, prefixes=''):
    """Prefix-stripped, shape-checked, non-strict checkpoint load (reference load.py:117-166)."""
    ckpt = read_checkpoint(path)
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
