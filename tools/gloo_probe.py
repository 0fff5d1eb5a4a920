import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    torch.cuda.set_device(0)
    out = []
    # 1. plain async all_reduce on a slice of a big buffer after a kernel writes it
    buf = torch.zeros(1 << 24, device="cuda")
    works = []
    for i in range(8):
        sl = buf[i * (1 << 21):(i + 1) * (1 << 21)]
        big = torch.randn(4096, 4096, device="cuda"); big = big @ big          # keep the stream busy
        sl.fill_(float(rank + 1) * (i + 1))
        works.append(dist.all_reduce(sl, async_op=True))
    for wk in works: wk.wait()
    torch.cuda.synchronize()
    exp = torch.cat([torch.full((1 << 21,), 3.0 * (i + 1)) for i in range(8)])
    out.append(float((buf.cpu() - exp).abs().max()))
    q.put((rank, out))
    dist.destroy_process_group()
if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=w, args=(r, 29571, q)) for r in range(2)]
    [p.start() for p in ps]; print([q.get(timeout=120) for _ in ps]); [p.join() for p in ps]
