"""GPU (-m gpu): a checkpoint written in the reference layout resumes training bit for bit in
its state (parameters, Adam moments, step count, scheduler) and within Adam's noise bound one step later."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _wrapper(resume=None):
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    from mindtheedge_amd.utils.config import load_config
    cfg = load_config(None, {"model": {"loss": {"supervised_method": "sparse-silog", "supervised_num_scales": 1, "supervised_loss_weight": 1.0,
                                                  "edges_depth_edge_loss_all_scales": True, "flip_lr_prob": 0.0},
                                       "depth_net": {"dropout": 0.0}, "scheduler": {"step_size": 1, "gamma": 0.5}}})
    w = ModelWrapper(cfg, resume=resume).cuda()
    w.configure_optimizers()
    return w


def _step(w, batch):
    w.optimizer.zero_grad()
    w.training_step(dict(batch))["loss"].sum().backward()
    w.optimizer.step()


def test_save_resume_roundtrip(tmp_path):
    from mindtheedge_amd import kernels as K
    from mindtheedge_amd.models.model_checkpoint import save_checkpoint, load_checkpoint
    from mindtheedge_amd.utils.synthetic import synthetic_batch
    K.set_compute_dtype("fp32")
    torch.manual_seed(1)
    a = _wrapper()
    a.train()
    batch = synthetic_batch(1, 64, 128, seed=5, device=torch.device("cuda", 0))
    for _ in range(2):
        _step(a, batch)
    # the reference writes the file when epoch 0 has finished, BEFORE the counter and the scheduler advance
    # (trainers/common_trainer.py:80-91): 'epoch' = 0, and a resume continues at epoch 1
    a.current_epoch = 0
    path = save_checkpoint(os.path.join(tmp_path, "run", "epoch=0.ckpt"), a)
    a.current_epoch += 1
    a.scheduler.step()
    ckpt = load_checkpoint(path)
    assert set(ckpt) == {"config", "epoch", "state_dict", "optimizer", "scheduler"}
    assert all(k.startswith("model.depth_net.") for k in ckpt["state_dict"]) and len(ckpt["state_dict"]) == 218
    assert set(ckpt["optimizer"]) == {"state", "param_groups"} and len(ckpt["optimizer"]["state"]) == 218
    assert ckpt["optimizer"]["param_groups"][0]["name"] == "Depth" and ckpt["epoch"] == 0
    # indices are the reference's depth_net.parameters() numbering: 216 dense + 70 sparse-branch slots + weight, bias
    assert len(ckpt["optimizer"]["param_groups"][0]["params"]) == 218 + 70

    b = _wrapper(resume=ckpt)
    b.train()
    assert b.current_epoch == 1 and b.optimizer.steps == 2
    b.scheduler.step()                                   # the trainer's scheduler step of the finished epoch
    assert b.optimizer.param_groups[0]["lr"] == a.optimizer.param_groups[0]["lr"] == 0.0002 * 0.5
    sa, sb = a.state_dict(), b.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    # moments live at different flat offsets only if the layout changed; compare per parameter through the torch layout
    oa, ob = a.optimizer.state_dict(), b.optimizer.state_dict()
    assert all(torch.equal(oa["state"][i]["exp_avg_sq"], ob["state"][i]["exp_avg_sq"]) for i in oa["state"])
    _step(a, batch)
    _step(b, batch)
    torch.cuda.synchronize()
    worst = max(float((sa[k] - sb[k]).abs().max()) for k in sa)          # same state, same batch: only split-K / atomics order differs
    assert worst <= 2.1e-4, worst                                        # an Adam step moves a parameter by ~lr at most: two runs differ by <= 2 lr


def test_depth_net_checkpoint_path_loads_reference_prefixes(tmp_path):
    """config.model.depth_net.checkpoint_path (setup_depth_net, reference model_wrapper.py:561-586) with the
    'model.depth_net.' prefix the reference's own checkpoints carry."""
    from mindtheedge_amd.models.model_wrapper import ModelWrapper
    from mindtheedge_amd.utils.config import load_config
    from oracle import packnet_oracle as po
    P = po.fixture_params()
    path = os.path.join(tmp_path, "tri.ckpt")
    torch.save({"state_dict": {"model.depth_net." + k: v for k, v in P.items()}}, path)
    cfg = load_config(None, {"model": {"depth_net": {"checkpoint_path": path, "dropout": 0.0},
                                       "loss": {"supervised_method": "sparse-silog", "supervised_num_scales": 1, "supervised_loss_weight": 1.0,
                                                "edges_depth_edge_loss_all_scales": True}}})
    w = ModelWrapper(cfg)
    sd = w.depth_net.state_dict()
    assert all(torch.equal(sd[k], P[k]) for k in P)
