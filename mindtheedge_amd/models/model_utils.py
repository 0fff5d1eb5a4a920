"""Batch helpers of the model layer (reference: packnet_sfm/models/model_utils.py:33-151)."""
import torch

_FLIP_INPUT_KEYS = ('rgb', 'rgb_context', 'input_depth', 'input_depth_context', 'rgb_edge')
_FLIP_OUTPUT_KEYS = ('uncertainty', 'inv_depths', 'inv_depths_context', 'inv_depths_rgbd')


def flip_lr(t):
    assert t.dim() == 4, 'You need to provide a [B,C,H,W] image to flip'
    return torch.flip(t, [3])


def _map_nested(x, fn):
    if isinstance(x, (list, tuple)):
        return [_map_nested(v, fn) for v in x]
    return fn(x)


def flip_batch_input(batch):
    out = dict(batch)
    for key in _FLIP_INPUT_KEYS:
        if key in out and torch.is_tensor(out[key]):
            out[key] = flip_lr(out[key])
    return out


def flip_output(output):
    out = dict(output)
    for key in _FLIP_OUTPUT_KEYS:
        if key in out:
            out[key] = _map_nested(out[key], flip_lr)
    return out


def merge_outputs(*outputs):
    merged = {'metrics': {}}
    for output in outputs:
        for key, val in output.items():
            if key == 'metrics':
                for k, v in val.items():
                    assert k not in merged['metrics'], 'Combining duplicated key {} to {}'.format(k, key)
                    merged['metrics'][k] = v
            elif key != 'loss':
                assert key not in merged, 'Adding duplicated key {}'.format(key)
                merged[key] = val
    return merged


def stack_batch(batch):
    if batch['rgb'].dim() == 5:
        assert batch['rgb'].shape[0] == 1, 'Only batch size 1 is supported for multi-cameras'
        for key, val in batch.items():
            if isinstance(val, list):
                batch[key] = [v[0] if torch.is_tensor(v) else v for v in val]
            else:
                batch[key] = val[0]
    return batch
