"""CPU: checkpoint format compatibility (checkpoint half) -- the reference stores
{'config','epoch','state_dict','optimizer','scheduler'} with torch.optim.Adam's optimizer layout
(packnet_sfm/models/model_checkpoint.py:71-81) and reads it back through utils/load.py:117-201."""
import os
import sys
import types

import pytest
import torch
import torch.nn as nn

from mindtheedge_amd.trainers.data_parallel import FlatParameters, FusedAdam
from mindtheedge_amd.utils.load import load_network, read_checkpoint


def _toy():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 5, 3), nn.GroupNorm(1, 5), nn.Conv2d(5, 2, 1))


def test_fused_adam_reads_and_writes_torch_adam_state():
    ref = _toy()
    opt = torch.optim.Adam([{'params': list(ref.parameters()), 'name': 'Depth', 'lr': 2e-4}])
    x = torch.randn(2, 3, 8, 8)
    for _ in range(3):
        opt.zero_grad()
        ref(x).square().mean().backward()
        opt.step()
    sd = opt.state_dict()

    mine = _toy()
    mine.load_state_dict(ref.state_dict())
    flat = FlatParameters(mine.parameters())
    fused = FusedAdam(flat, lr=1e-4)
    fused.load_state_dict(sd)
    assert fused.steps == 3 and fused.param_groups[0]['lr'] == 2e-4
    for i, p in enumerate(mine.parameters()):                       # natural order = torch's state index
        o = flat.offsets[[id(q) for q in flat.params].index(id(p))]
        assert torch.equal(fused.exp_avg[o:o + p.numel()].view(p.shape), sd['state'][i]['exp_avg'])
        assert torch.equal(fused.exp_avg_sq[o:o + p.numel()].view(p.shape), sd['state'][i]['exp_avg_sq'])

    out = fused.state_dict()                                        # ...and back into a fresh torch.optim.Adam
    fresh = torch.optim.Adam([{'params': list(_toy().parameters()), 'name': 'Depth'}])
    fresh.load_state_dict(out)
    back = fresh.state_dict()
    assert back['param_groups'][0]['lr'] == 2e-4 and back['param_groups'][0]['name'] == 'Depth'
    for i in sd['state']:
        assert torch.equal(back['state'][i]['exp_avg'], sd['state'][i]['exp_avg'])
        assert float(back['state'][i]['step']) == 3.0
    with pytest.raises(ValueError):
        bad = {'state': {}, 'param_groups': [{'params': [0, 1], 'name': 'Depth'}]}
        fused.load_state_dict(bad)


def test_state_before_the_first_step_is_empty_like_torch():
    flat = FlatParameters(_toy().parameters())
    sd = FusedAdam(flat).state_dict()
    assert sd['state'] == {} and sd['param_groups'][0]['params'] == list(range(6))


def test_reference_style_checkpoint_with_foreign_config_class(tmp_path):
    """A checkpoint whose 'config' is a yacs CfgNode (not installed here) and whose tensors carry the reference's
    'model.depth_net.' prefix plus tensors this build does not have (sparse branch): the network part still loads."""
    yacs = types.ModuleType('yacs')
    cfgmod = types.ModuleType('yacs.config')
    CfgNode = type('CfgNode', (dict,), {'__module__': 'yacs.config', '__qualname__': 'CfgNode'})
    cfgmod.CfgNode = CfgNode
    yacs.config = cfgmod
    sys.modules['yacs'], sys.modules['yacs.config'] = yacs, cfgmod
    try:
        src = _toy()
        sd = {'model.depth_net.' + k: v for k, v in src.state_dict().items()}
        sd['model.depth_net.mconvs.0.layer1.0.kernel'] = torch.zeros(25, 1, 32)          # sparse-branch tensor: ignored
        sd['model.depth_net.0.weight'] = src.state_dict()['0.weight']
        path = os.path.join(tmp_path, 'ref.ckpt')
        torch.save({'config': CfgNode(arch=CfgNode(seed=42)), 'epoch': 7, 'state_dict': sd}, path)
    finally:
        del sys.modules['yacs'], sys.modules['yacs.config']
    ckpt = read_checkpoint(path)
    assert ckpt['epoch'] == 7 and ckpt['config']['arch']['seed'] == 42
    dst = _toy()
    for p in dst.parameters():
        p.data.zero_()
    load_network(dst, path, ['depth_net', 'disp_network'])
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v)
