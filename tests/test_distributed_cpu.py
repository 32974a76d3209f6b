"""CPU, world_size 2 over gloo: the data-parallel glue (one flat gradient arena per network,
one all-reduce each, weights broadcast from rank 0) -- the same code path that runs over RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


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
        from dfmir_amd import distributed as D
        from dfmir_amd.optim import FlatAdam
        torch.manual_seed(100 + rank)              # ranks start from DIFFERENT weights
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
        opt = FlatAdam(net.parameters(), lr=2e-4, betas=(0.5, 0.999))
        assert D.is_distributed() and D.world_size() == world
        D.broadcast_arena(opt.flat_p, src=0)
        opt.grad_scale = 1.0 / D.world_size()
        w0 = opt.flat_p.clone()
        x = torch.full((4, 5), float(rank + 1))    # rank-dependent batch shard
        opt.zero_grad()
        net(x).pow(2).sum().backward()
        local = opt.flat_g.clone()
        D.allreduce_arenas([opt.flat_g])
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        avg = D.allreduce_scalars([float(rank)])
        q.put((rank, w0.numpy().copy(), opt.flat_g.numpy().copy(), sum(gathered).numpy().copy(), avg))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_arena_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0a, ga, sa, avga), (_, w0b, gb, sb, avgb) = res
    import numpy as np
    assert np.array_equal(w0a, w0b), "weights must be identical after the rank-0 broadcast"
    assert np.allclose(ga, sa) and np.allclose(gb, sb) and np.array_equal(ga, gb)
    assert float(np.abs(ga).sum()) > 0
    assert avga == avgb == [0.5]
