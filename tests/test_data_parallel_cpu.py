"""CPU, world_size 2, gloo: the bucketed gradient all-reduce that bench.py / the trainer use over RCCL.
Checks (a) flat re-homing keeps parameters/gradients aliased, (b) buckets cover the buffer in reverse execution order,
(c) the overlapped all-reduce launched from grad-ready hooks yields exactly the average of the per-rank gradients
(= what hvd.DistributedOptimizer would have produced, horovod_trainer.py:53-55), including parameters that received no
gradient on this step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mindtheedge_amd.trainers.data_parallel import FlatParameters, BucketedAllReduce, broadcast_parameters
        torch.manual_seed(100 + rank)                                  # ranks start different on purpose
        net = nn.Sequential(nn.Linear(7, 33), nn.Tanh(), nn.Linear(33, 65), nn.Tanh(), nn.Linear(65, 3))
        unused = nn.Parameter(torch.ones(5))
        params = list(net.parameters()) + [unused]
        flat = FlatParameters(params)
        broadcast_parameters(flat)
        ref0 = [torch.zeros_like(flat.flat) for _ in range(world)]
        dist.all_gather(ref0, flat.flat)
        assert all(torch.equal(r, ref0[0]) for r in ref0), "broadcast must equalise the ranks"
        for p, o in zip(flat.params, flat.offsets):
            assert p.data_ptr() == flat.flat.data_ptr() + 4 * o and p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o
        red = BucketedAllReduce(flat, bucket_bytes=4 * 200)            # several buckets
        assert len(red.buckets) >= 3 and red.buckets[0][0] == 0 and red.buckets[-1][1] == flat.total
        assert flat.params[0] is unused and flat.params[1] is params[-2]      # reverse execution order
        for step in range(2):
            flat.zero_grad()
            x = torch.randn(4 + rank, 7, generator=torch.Generator().manual_seed(10 * step + rank))
            net(x).square().mean().backward()
            local = flat.grad.clone()                                  # hooks already launched the collectives: recompute
            scale = red.finish()
            assert scale == 1.0 / world
            # reference: gather every rank's purely local gradient
            flat2 = [p.detach().clone().requires_grad_(True) for p in params[:-1]]
            net2 = nn.Sequential(nn.Linear(7, 33), nn.Tanh(), nn.Linear(33, 65), nn.Tanh(), nn.Linear(65, 3))
            for p2, p in zip(net2.parameters(), params[:-1]):
                p2.data.copy_(p.data)
            net2(x).square().mean().backward()
            mine = torch.zeros_like(flat.grad)
            for p, o in zip(flat.params, flat.offsets):
                src = dict(zip([id(q_) for q_ in params[:-1]], net2.parameters())).get(id(p))
                if src is not None:
                    mine[o:o + p.numel()] = src.grad.flatten()
            allg = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allg, mine)
            want = sum(allg) / world
            assert torch.allclose(flat.grad * scale, want, rtol=1e-6, atol=1e-7), float((flat.grad * scale - want).abs().max())
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res
