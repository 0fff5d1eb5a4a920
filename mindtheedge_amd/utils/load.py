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


class _AllowListPickle:
    """pickle module for torch.load with an ALLOW-LIST unpickler: tensors / storages / plain containers / numpy scalars, and
    the reference's yacs ``CfgNode`` (its 'config' entry) mapped to a dict stand-in, so yacs is not needed.  Any other
    global raises UnpicklingError: an untrusted .ckpt cannot name code to run."""
    import pickle as _p
    __name__ = 'pickle'
    load, loads, dump, dumps = _p.load, _p.loads, _p.dump, _p.dumps
    PickleError, UnpicklingError, Pickler = _p.PickleError, _p.UnpicklingError, _p.Pickler
    _OK = {('collections', 'OrderedDict'), ('torch._utils', '_rebuild_tensor_v2'), ('torch._utils', '_rebuild_parameter'),
           ('torch', 'Size'), ('torch', 'device'), ('torch.serialization', '_get_layout'), ('_codecs', 'encode'),
           ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'), ('numpy', 'dtype'),
           ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'), ('numpy', 'ndarray'),
           ('builtins', 'set'), ('builtins', 'frozenset'), ('builtins', 'slice'), ('builtins', 'complex')}

    class Unpickler(_p.Unpickler):
        def find_class(self, module, name):
            if module.split('.')[0] == 'yacs' and name == 'CfgNode':
                return _CfgStub
            if (module, name) in _AllowListPickle._OK or (module == 'torch' and (name.endswith('Storage') or name in _TORCH_DTYPES)):
                return super().find_class(module, name)
            raise _AllowListPickle._p.UnpicklingError("checkpoint names the global %s.%s, which is not on the allow-list "
                                                      "(pass unsafe=True / MTE_UNSAFE_CKPT=1 only for a file you trust)" % (module, name))


_TORCH_DTYPES = ('float32', 'float64', 'float16', 'bfloat16', 'int64', 'int32', 'int16', 'int8', 'uint8', 'bool')


def read_checkpoint(path, unsafe=False):
    """torch.load of a `.ckpt` written by this package or by the reference (models/model_checkpoint.py:71-81).

    First the weights-only unpickler; a file it rejects because of a non-tensor class (the reference's 'config' entry is a
    yacs CfgNode) is read again through an allow-list unpickler (_AllowListPickle).  Anything else in the file raises, and
    a truncated / corrupt file is reported, not re-parsed permissively.  ``unsafe=True`` (or MTE_UNSAFE_CKPT=1) opts into
    the full unpickler."""
    import os
    import pickle
    import torch
    if unsafe or os.environ.get('MTE_UNSAFE_CKPT') == '1':
        return torch.load(path, map_location='cpu', weights_only=False)
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError:
        return torch.load(path, map_location='cpu', weights_only=False, pickle_module=_AllowListPickle)


def backwards_state_dict(state_dict):
    """Key names of an old-format (`.pth.tar`) model file in today's layout (reference load.py:169-201): every key gains the
    `model.` prefix, `model.model.` collapses to `model.`, `pose_network` / `disp_network` become `pose_net` / `depth_net`, and the
    depth network's `conv3.0.weight` / `conv3.0.bias` lose the Sequential index."""
    renames = (('model.model.', 'model.'), ('pose_network.', 'pose_net.'), ('disp_network.', 'depth_net.'))
    out = type(state_dict)() if hasattr(state_dict, 'items') else {}
    for key, val in state_dict.items():
        key = 'model.' + key
        if 'disp_network' in key:
            key = key.replace('conv3.0.weight', 'conv3.weight').replace('conv3.0.bias', 'conv3.bias')
        for old, new in renames:
            key = key.replace(old, new)
        out[key] = val
    return out


def load_network(network, path, prefixes=''):
    """Prefix-stripped, shape-checked, non-strict checkpoint load (reference load.py:117-166).

    As upstream, a prefix matches ANYWHERE in a key (`prefix + '.' in key`: `module.model.depth_net.conv1.weight` loads with
    prefix `depth_net`) and everything up to and including its first occurrence is cut; the cut key is what the next prefix of the
    list is tested against.  A `.pth.tar` path goes through backwards_state_dict first."""
    prefixes = make_list(prefixes)
    if isinstance(path, str):
        ckpt = read_checkpoint(path)
        sd = ckpt.get('state_dict', ckpt)
        if path.endswith('.pth.tar'):
            sd = backwards_state_dict(sd)
    else:                                        # a state dict (reference :139-140: the resume path passes one)
        sd = path
    own = network.state_dict()
    picked = {}
    for key, val in sd.items():
        for prefix in prefixes:
            p = prefix + '.'
            if p in key:
                key = key[key.find(p) + len(p):]
                if key in own and tuple(own[key].shape) == tuple(val.shape):
                    picked[key] = val
    network.load_state_dict(picked, strict=False)
    return network
