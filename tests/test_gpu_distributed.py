"""GPU, world_size 2 on ONE device over gloo (device tensors staged through the host by dfmir_amd.distributed): the
multi-process path of the REAL models -- REGISTRATIONModel.parallelize() / sync_gradients(), Registration3DModel, and
bench.py launched exactly as the driver launches it (torch.distributed.run, one rank per process).  RCCL itself needs
one GPU per rank; what runs here is every line around the collective."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_2d(rank, world, port, q, capture):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DFMIR_DIST_BACKEND="gloo")
    sys.path.insert(0, REPO)
    from dfmir_amd import distributed as D
    from dfmir_amd import ops
    from dfmir_amd.options import default_options
    from dfmir_amd.registration_model import REGISTRATIONModel
    from tests.golden import common as C
    D.init_from_env()
    try:
        size, B = 64, 2
        opt = default_options(batch_size=B, crop_size=size, load_size=size, ngf=8, gpu_ids=[0],
                              checkpoints_dir="/tmp/dfmir_ddp", name="r%d" % rank, capture_step=capture)
        torch.manual_seed(100 + rank)                      # ranks start from DIFFERENT weights
        model = REGISTRATIONModel(opt)
        with torch.no_grad():
            model.netR.flow.weight.mul_(1e4)
        ops.seed_patch_ids(77, "cuda")                     # same ids on both ranks: gradients comparable below
        A0, B0 = C.image_pair(300 + 10 * rank, B, size, size)    # rank-dependent batch shard
        data = {"A": A0.cuda(), "B": B0.cuda(), "A_paths": [""] * B, "B_paths": [""] * B}
        model.data_dependent_initialize(data)
        model.setup(opt)
        model.parallelize()
        assert model._ddp and all(o.grad_scale == 0.5 for o in model.optimizers)
        w_after_bcast = [o.flat_p.cpu() for o in model.optimizers]
        # one step by hand: local gradients, then the exchange
        model.set_input(data)
        model._forward_backward()
        local = [o.flat_g.cpu() for o in model.optimizers]
        model.sync_gradients()
        summed = [o.flat_g.cpu() for o in model.optimizers]
        gathered = []
        for t in local:
            parts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            gathered.append(sum(parts))
        for o in model.optimizers:
            o.step()
        # then the public entry point for a few more steps (graph capture included when asked for)
        for it in range(4):
            model.set_input(data)
            model.optimize_parameters()
        losses = model.get_current_losses()
        w_end = [o.flat_p.cpu() for o in model.optimizers]
        graphed = bool(getattr(model, '_graph', {}).get('graph') is not None)
        q.put((rank, [w.numpy() for w in w_after_bcast], [s.numpy() for s in summed], [g.numpy() for g in gathered],
               [w.numpy() for w in w_end], dict(losses), graphed))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("capture", [False, True], ids=["eager", "graph"])
def test_registration_model_two_ranks(capture):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_2d, args=(r, 2, port, q, capture)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0, r1 = res
    for a, b in zip(r0[1], r1[1]):
        assert np.array_equal(a, b), "weights must be identical after the rank-0 broadcast in parallelize()"
    for s0, s1, g0 in zip(r0[2], r1[2], r0[3]):
        assert np.array_equal(s0, s1), "both ranks hold the same reduced arena"
        assert np.allclose(s0, g0, rtol=1e-6, atol=1e-12), "flat_g after sync_gradients = sum of the ranks' local gradients"
        assert float(np.abs(s0).sum()) > 0
    for a, b in zip(r0[4], r1[4]):
        assert np.array_equal(a, b), "replicas stay bit-identical through the optimizer steps"
    assert all(np.isfinite(v) for v in r0[5].values()) and all(np.isfinite(v) for v in r1[5].values())
    assert r0[5] != r1[5]                                   # different shards: per-rank losses differ
    assert r0[6] == r1[6] == capture


def _worker_3d(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DFMIR_DIST_BACKEND="gloo")
    sys.path.insert(0, REPO)
    from dfmir_amd import distributed as D
    from dfmir_amd.registration3d import Registration3DModel
    from tests.golden import common as C
    D.init_from_env()
    try:
        shape = (32, 32, 32)
        torch.manual_seed(200 + rank)
        m = Registration3DModel(shape, None, device="cuda")
        with torch.no_grad():
            m.netR.flow.weight.mul_(3e4)
        m.parallelize()
        w0 = m.optimizer_R.flat_p.cpu().numpy()
        A = C.rand(31 + rank, 1, 1, *shape).cuda()
        B = 0.5 * A + 0.5 * C.rand(41 + rank, 1, 1, *shape).cuda()
        for _ in range(3):
            m.set_input({"A": A, "B": B})
            m.optimize_parameters()
        q.put((rank, w0, m.optimizer_R.flat_p.cpu().numpy(), m.optimizer_R.flat_g.cpu().numpy(), m.get_current_losses()))
    finally:
        dist.destroy_process_group()


def test_registration3d_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_3d, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0, r1 = res
    assert np.array_equal(r0[1], r1[1]) and np.array_equal(r0[2], r1[2]) and np.array_equal(r0[3], r1[3])
    assert not np.array_equal(r0[1], r0[2])                 # the replicas trained
    assert np.isfinite(list(r0[4].values())).all() and r0[4] != r1[4]


def test_bench_two_ranks_as_the_driver_launches_it():
    """python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ... (the driver's N > 1 command line) on a
    small geometry, ranks sharing the one GPU over gloo: ONE JSON line from rank 0, whole-job throughput, n_gpus 2."""
    env = dict(os.environ, DFMIR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3",
           "--batch", "2", "--size", "64", "--ngf", "8"]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 4 and r["config"]["parallelism"] == "dp2"
    assert r["value"] > 0 and r["scaling"] == "weak" and r["step_submission"].startswith("hipGraph")
    assert abs(r["value"] - 4 * r["steps"] / (r["ms_per_step"] * 1e-3 * r["steps"])) < 1e-6 * r["value"]
    assert all(np.isfinite(v) for v in r["losses"].values())
