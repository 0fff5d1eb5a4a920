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


def _worker_real(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mindtheedge_amd.trainers.data_parallel import FlatParameters, BucketedAllReduce, broadcast_parameters
        from mindtheedge_amd.networks.depth.PackNetSAN01 import PackNetSAN01
        torch.manual_seed(5 + rank)
        net = PackNetSAN01(dropout=0.5, version="1A")                  # the real parameter set: 218 tensors, 77 M elements
        flat = FlatParameters(net.parameters())
        broadcast_parameters(flat)
        red = BucketedAllReduce(flat)                                  # defaults: 32 MB buckets, 32 MB messages
        assert red.active and red.world == world
        sizes = [(e - s) * 4 for s, e, _ in red.buckets]
        big = max(range(len(sizes)), key=sizes.__getitem__)
        assert sizes[big] > 150 << 20                                  # pack5.conv (151 MB) cannot be split across buckets ...
        msgs = red._messages(*red.buckets[big][:2])
        assert len(msgs) >= 5 and all((e - s) * 4 <= 32 << 20 for s, e in msgs)          # ... but goes out as <= 32 MB messages
        assert msgs[0][0] == red.buckets[big][0] and msgs[-1][1] == red.buckets[big][1]
        assert all(a[1] == b[0] for a, b in zip(msgs, msgs[1:]))       # contiguous, nothing dropped, nothing twice
        assert sum(sizes) == flat.total * 4 and sizes[-1] <= 2 << 20   # small exposed tail bucket
        d = red.describe()
        assert d["message_mb"] == 32.0 and len(d["buckets_mb"]) == len(red.buckets) and max(d["messages_per_bucket"]) == len(msgs)

        # "backward": every rank owns an independent gradient; parameters become ready in flat order (reverse execution order),
        # each readiness notification may complete a bucket and launch its chunked all-reduce, as the grad hooks do on the GPU
        idx = torch.arange(flat.total, dtype=torch.float32)
        grad_of = lambda r: torch.sin(idx * (1e-3 * (r + 1))) + 0.25 * r
        for step in range(2):
            flat.zero_grad()
            flat.grad.copy_(grad_of(rank) * (step + 1))
            for p in flat.params:
                if not getattr(p, "_mte_flat_tail", False):            # the SAN fusion scalars get no gradient on this path
                    red._on_grad_ready(p)
            launched = [b for b, _, _, _ in red.launch_log]
            assert launched == sorted(launched) and len(launched) >= len(red.buckets) - 1
            scale = red.finish()                                       # reduces the bucket the unused parameters kept open, waits
            want = sum(grad_of(r) for r in range(world)) / world * (step + 1)
            assert scale == 1.0 / world
            assert torch.allclose(flat.grad * scale, want, rtol=1e-6, atol=1e-6), float((flat.grad * scale - want).abs().max())
            assert red.describe()["launch_order"] == list(range(len(red.buckets)))
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_real_network_chunked_buckets_average_of_per_rank_gradients_world2_gloo():
    """DDP parity as SURVEY.md 8(e) defines it: the reduced gradient is the AVERAGE of the ranks' independent gradients
    (hvd.DistributedOptimizer, horovod_trainer.py:53-55), on the real network's 77 M-parameter layout with the 182 MB bucket
    cut into <= 32 MB all-reduce messages."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res
