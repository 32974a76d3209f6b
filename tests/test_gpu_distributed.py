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


def _reap(procs):
    """A worker that hangs (a collective that never completes) must not outlive its test holding the GPU."""
    for p in procs:
        if p.is_alive():
            p.terminate()
            p.join(timeout=20)
            if p.is_alive():
                p.kill()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_2d(rank, world, port, q, capture, backend="gloo", cfg=(64, 2, 8), bucket=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DFMIR_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, REPO)
    from dfmir_amd import distributed as D
    from dfmir_amd import ops
    from dfmir_amd.options import default_options
    from dfmir_amd.registration_model import REGISTRATIONModel
    from tests.golden import common as C
    D.init_from_env()
    try:
        size, B, ngf = cfg
        big = size * size * B * ngf > 64 * 64 * 2 * 8           # full geometry: arenas travel back as digests
        opt = default_options(batch_size=B, crop_size=size, load_size=size, ngf=ngf, gpu_ids=[torch.cuda.current_device()],
                              checkpoints_dir="/tmp/dfmir_ddp", name="r%d" % rank, capture_step=capture,
                              bucket_allreduce=bucket)
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
        assert model._ddp and all(o.grad_scale == 1.0 / world for o in model.optimizers)
        w_after_bcast = [o.flat_p.cpu() for o in model.optimizers]
        if bucket and not capture:                           # the arena's tail leaves INSIDE backward
            model.set_input(data)
            model._forward_backward()
            assert model._bucket['fired'] == 1 and len(model._early) == 1
            model.sync_gradients()                             # the arena's head and the other arenas
            summed = [o.flat_g.cpu().numpy() for o in model.optimizers]
            for o in model.optimizers:
                o.step()
            for it in range(4):
                model.set_input(data)
                model.optimize_parameters()
            assert model._bucket['fired'] == 5 and not model._early
            q.put((rank, [w.numpy() for w in w_after_bcast], summed, model._bucket['off'],
                   [o.flat_p.cpu().numpy() for o in model.optimizers], dict(model.get_current_losses()), False))
            return
        # one step by hand: local gradients, then the exchange (the patch ids this rank draws are recorded on the way)
        drawn = []
        orig_sets = model._patch_id_sets
        model._patch_id_sets = lambda *a, **k: (lambda out: (drawn.append([t.cpu().numpy() for t in out]), out)[1])(orig_sets(*a, **k))
        model.set_input(data)
        model._forward_backward()
        del model._patch_id_sets                                 # (back to the class's method before anything is captured)
        local = [o.flat_g.cpu() for o in model.optimizers]
        model.sync_gradients()
        summed = [o.flat_g.cpu() for o in model.optimizers]
        gathered = []
        for t in local:
            t_ = t.cuda() if backend == "nccl" else t
            parts = [torch.zeros_like(t_) for _ in range(world)]
            dist.all_gather(parts, t_)
            gathered.append(sum(parts).cpu())
        for o in model.optimizers:
            o.step()
        # then the public entry point for a few more steps (graph capture included when asked for)
        for it in range(4):
            model.set_input(data)
            model.optimize_parameters()
        losses = model.get_current_losses()
        w_end = [o.flat_p.cpu() for o in model.optimizers]
        graphed = bool(getattr(model, '_graph', {}).get('graph') is not None)
        if bucket:                                           # a capturing model keeps the whole exchange behind the replay
            assert model._bucket is not None and model._bucket['fired'] == 0 and not model._early
        if big:
            import hashlib
            dig = lambda ts: [hashlib.sha1(t.numpy().tobytes()).hexdigest() for t in ts]
            sum_ok = [bool(np.allclose(s_.numpy(), g_.numpy(), rtol=1e-6, atol=1e-12)) and float(s_.abs().sum()) > 0
                      for s_, g_ in zip(summed, gathered)]
            q.put((rank, dig(w_after_bcast), dig(summed), sum_ok, dig(w_end), dict(losses), graphed, drawn))
        else:
            q.put((rank, [w.numpy() for w in w_after_bcast], [s.numpy() for s in summed], [g.numpy() for g in gathered],
                   [w.numpy() for w in w_end], dict(losses), graphed, drawn))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("capture", [False, True], ids=["eager", "graph"])
def test_registration_model_two_ranks(capture):
    r0, r1 = _run_two_ranks_2d(capture, "gloo")
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


def test_gradient_buckets_change_nothing():
    """opt.bucket_allreduce: G's late layers (modules 17.., final when the main pass's backward has reached module 16)
    are flushed and all-reduced from INSIDE backward, the rest of the arena after it.  Five steps on two ranks end with
    bit-identical replicas, and with the weights of the one-exchange-per-arena protocol."""
    b0, b1 = _run_two_ranks_2d(False, "gloo", bucket=True)
    p0, p1 = _run_two_ranks_2d(False, "gloo")
    assert 0 < b0[3] == b1[3]                                # the bucket boundary inside G's arena
    for a, b in zip(b0[4], b1[4]):
        assert np.array_equal(a, b), "replicas stay bit-identical with the early bucket"
    # against a separate run of the one-exchange protocol: the reduced gradient arenas of the first step agree to the
    # run-to-run noise of the split-K weight-gradient atomics (head AND tail of G's arena), the weights after five Adam
    # steps within the movement that noise can cause (an entry with a near-zero gradient moves by +-lr per step)
    for a, b0_, b1_ in zip(p0[2], b0[2], b1[2]):
        assert np.array_equal(b0_, b1_), "both ranks hold the same reduced arena"
        assert float(np.linalg.norm(a - b0_)) <= 1e-4 * float(np.linalg.norm(a)), float(np.linalg.norm(a - b0_) / np.linalg.norm(a))
    off = b0[3]
    g_plain, g_bucket = p0[2][0], b0[2][0]
    for sl in (slice(0, off), slice(off, None)):
        assert float(np.linalg.norm(g_plain[sl] - g_bucket[sl])) <= 1e-4 * float(np.linalg.norm(g_plain[sl]))
    for a, b in zip(b0[4], p0[4]):
        assert float(np.abs(a - b).max()) <= 2 * 5 * 2e-4 and float(np.linalg.norm(a - b)) <= 5e-3 * float(np.linalg.norm(b))
    for k in b0[5]:
        assert abs(b0[5][k] - p0[5][k]) <= 2e-2 * max(abs(p0[5][k]), 1e-6), (k, b0[5][k], p0[5][k])


def test_gradient_buckets_with_a_captured_step():
    """opt.bucket_allreduce together with opt.capture_step (what `bench.py --bucket-allreduce` runs): the depth-bucket hook
    stays off for a model that captures its step -- in the eager warm-up steps too, so the deferred-gradient job table the
    capture needs exists before it (no table upload inside the capture) and every rank issues the same collectives whether
    or not its capture succeeded.  Two ranks: the later steps ARE graph replays, the hook never fired, the reduced arenas
    are the sums of the local gradients and the replicas stay bit-identical."""
    r0, r1 = _run_two_ranks_2d(True, "gloo", bucket=True)
    for s0, s1, g0 in zip(r0[2], r1[2], r0[3]):
        assert np.array_equal(s0, s1) and np.allclose(s0, g0, rtol=1e-6, atol=1e-12) and float(np.abs(s0).sum()) > 0
    for a, b in zip(r0[4], r1[4]):
        assert np.array_equal(a, b), "replicas stay bit-identical through the optimizer steps"
    assert r0[6] and r1[6], "the later steps must have been hipGraph replays (the capture did not fall back)"


@pytest.mark.parametrize("capture,bucket", [(False, False), (True, False), (True, True)], ids=["eager", "graph", "graph+bucket-option"])
def test_registration_model_eight_ranks_rehearsal(capture, bucket):
    """The 8-GPU protocol of BASELINE configs[2] rehearsed with EIGHT processes on the one GPU (gloo, device tensors staged
    through the host; 64 x 64, ngf 8, batch 2 per rank): every rank holds rank 0's weights after parallelize(), draws the
    SAME patch ids, flat_g after sync_gradients() is the sum of the eight local gradients on every rank, and after the
    manual step + four optimize_parameters() calls (eager, or captured + replayed, with and without opt.bucket_allreduce)
    the eight replicas are bit-identical.  What the first SCALE run on a real 8-GPU node adds is RCCL itself."""
    res = _run_two_ranks_2d(capture, "gloo", bucket=bucket, world=8)
    assert [r[0] for r in res] == list(range(8))
    r0 = res[0]
    assert r0[7] and all(len(r[7]) == len(r0[7]) for r in res)
    for r in res[1:]:
        for a, b in zip(r0[1], r[1]):
            assert np.array_equal(a, b), "weights must be identical after the rank-0 broadcast in parallelize()"
        for s0, s1 in zip(r0[2], r[2]):
            assert np.array_equal(s0, s1), "every rank holds the same reduced arena"
        for a, b in zip(r0[4], r[4]):
            assert np.array_equal(a, b), "replicas stay bit-identical through the optimizer steps"
        for call0, call in zip(r0[7], r[7]):
            for i0, i1 in zip(call0, call):
                assert np.array_equal(i0, i1), "every rank draws the same patch ids (same generator seed)"
        assert r[6] == capture
    for s0, g0 in zip(r0[2], r0[3]):
        # (eight addends: the ring's summation order is not the test's; norm-wise)
        assert float(np.linalg.norm(s0 - g0)) <= 1e-6 * float(np.linalg.norm(g0)) and float(np.abs(s0).sum()) > 0, \
            "flat_g = sum of the 8 local gradients"
    assert len({tuple(sorted(r[5].items())) for r in res}) == 8, "eight different shards: eight different per-rank losses"


def test_bench_eight_ranks_as_the_driver_launches_it():
    """The driver's N = 8 command line (torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 ...) on a small geometry,
    the eight ranks sharing the one GPU over gloo: ONE JSON line, n_gpus 8, global batch 16, the all-reduce of ones saw eight
    ranks, eight per-rank step times."""
    env = dict(os.environ, DFMIR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "3",
           "--batch", "2", "--size", "64", "--ngf", "8"]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["config"]["global_batch"] == 16 and r["config"]["parallelism"] == "dp8"
    assert r["collective"]["rccl_ranks_seen"] == 8 and len(r["collective"]["step_ms_by_rank"]) == 8
    assert all(t > 0 for t in r["collective"]["step_ms_by_rank"])
    assert r["value"] > 0 and r["scaling"] == "weak" and r["step_submission"].startswith("hipGraph")
    assert all(np.isfinite(v) for v in r["losses"].values())


def test_registration_model_two_ranks_full_geometry():
    """BASELINE configs[2]'s per-GPU shard as quoted -- batch 16 per rank, 256 x 256, ngf 64, the step captured into a
    hipGraph -- on two ranks (sharing the one GPU over gloo): identical weights after the broadcast, flat_g after the
    exchange = the sum of the ranks' local gradients, both ranks hold the same reduced arenas, replicas bit-identical
    after the eager steps and the graph replays (arenas compared by SHA-1: 45 MB each)."""
    r0, r1 = _run_two_ranks_2d(True, "gloo", cfg=(256, 16, 64))
    assert r0[1] == r1[1], "weights must be identical after the rank-0 broadcast in parallelize()"
    assert r0[2] == r1[2], "both ranks hold the same reduced arena"
    assert all(r0[3]) and all(r1[3]), "flat_g after sync_gradients = sum of the ranks' local gradients"
    assert r0[4] == r1[4], "replicas stay bit-identical through the optimizer steps"
    assert all(np.isfinite(v) for v in r0[5].values()) and r0[5] != r1[5]
    assert r0[6] and r1[6], "the later steps must have been hipGraph replays"


def _run_two_ranks_2d(capture, backend, cfg=(64, 2, 8), bucket=False, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_2d, args=(r, world, port, q, capture, backend, cfg, bucket)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=900) for _ in procs], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        _reap(procs)
    return res


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (auto-runs on a >= 2-GPU box)")
@pytest.mark.parametrize("capture", [False, True], ids=["eager", "graph"])
def test_registration_model_two_gpus_rccl(capture):
    """The same protocol over the REAL backend: init_process_group("nccl", device_id=...), weight broadcast, local
    backward (eager, then captured + replayed), async per-arena all-reduce, Adam -- one rank per GPU."""
    r0, r1 = _run_two_ranks_2d(capture, "nccl")
    for a, b in zip(r0[1], r1[1]):
        assert np.array_equal(a, b)
    for s0, s1, g0 in zip(r0[2], r1[2], r0[3]):
        assert np.array_equal(s0, s1)
        assert np.allclose(s0, g0, rtol=1e-5, atol=1e-10)
    for a, b in zip(r0[4], r1[4]):
        assert np.array_equal(a, b), "replicas stay bit-identical through the optimizer steps"
    assert r0[5] != r1[5] and r0[6] == r1[6] == capture


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (auto-runs on a >= 2-GPU box)")
def test_bench_two_gpus_rccl():
    """bench.py at the driver's N = 2 command line over RCCL, full geometry: the all-reduce payload is the 49.1 MB of
    BASELINE.md section 3 and the exposed collective time is reported."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DFMIR_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "4"]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 32
    assert abs(r["collective"]["payload_bytes"] - 49.1e6) < 0.2e6
    assert r["collective"]["backend"] == "nccl" and r["collective"]["exposed_ms_per_step"] >= 0.0


def test_bench_single_rank_over_rccl():
    """bench.py with a process group of ONE rank over the real backend (DFMIR_FORCE_DIST=1): RCCL communicator created
    with device_id, weights broadcast, the step captured into a hipGraph (thread-local capture mode, RCCL's watchdog
    thread alive), the three arena all-reduces issued asynchronously through RCCL after every replay, exposure timed.
    Everything the 8-GPU launch does except talking to a second GPU."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DFMIR_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.pop("DFMIR_DIST_BACKEND", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "4", "--batch", "2",
           "--size", "64", "--ngf", "8", "--no-3d", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["step_submission"].startswith("hipGraph"), r["step_submission"]
    assert r["collective"]["backend"] == "nccl" and r["collective"]["exposed_ms_per_step"] >= 0.0
    assert r["collective"]["payload_bytes"] > 0 and all(np.isfinite(v) for v in r["losses"].values())


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
    try:
        res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        _reap(procs)
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


def _gmn_data():
    """Four pairs whose mask sums differ strongly between the two halves of the batch (images 2, 3: top 60 % background)."""
    from tests.golden import common as C
    A, B = C.image_pair(300, 4, 64, 64)
    A[2:, :, :38], B[2:, :, :38] = -1.0, -1.0
    return A, B


def _gmn_model(B, rank_tag, capture=False, gmn=False):
    from dfmir_amd.options import default_options
    from dfmir_amd.registration_model import REGISTRATIONModel
    from tests.test_gpu_models import PinnedIds
    opt = default_options(batch_size=B, crop_size=64, load_size=64, ngf=8, gpu_ids=[torch.cuda.current_device()],
                          checkpoints_dir="/tmp/dfmir_ddp", name="gmn%s" % rank_tag, capture_step=capture,
                          global_mask_norm=gmn)
    torch.manual_seed(321)
    model = REGISTRATIONModel(opt)
    with torch.no_grad():
        model.netR.flow.weight.mul_(1e4)
        # a random-init G outputs ~0 everywhere (> -0.95: every mask pixel set, equal mask sums on all shards); push its
        # output to tanh(-3) = -0.995 so that the masks follow the data: (real_B > -0.95) | (registered > -0.95)
        dict(model.netG.named_parameters())['model.30.bias'].fill_(-3.0)
    model.patch_id_source = PinnedIds("cuda")
    return model, opt


def _worker_gmn(rank, world, port, q, gmn):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DFMIR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, REPO)
    from dfmir_amd import distributed as D
    D.init_from_env()
    try:
        A, B = _gmn_data()
        model, opt = _gmn_model(2, "r%d" % rank, gmn=gmn)
        # the data-dependent init on the SAME pair everywhere (it only creates netF; rank 0's weights are broadcast)
        init = {"A": A[:2].cuda(), "B": B[:2].cuda(), "A_paths": [""] * 2, "B_paths": [""] * 2}
        model.data_dependent_initialize(init)
        model.setup(opt)
        model.parallelize()
        sl = slice(2 * rank, 2 * rank + 2)
        model.set_input({"A": A[sl].cuda(), "B": B[sl].cuda(), "A_paths": [""] * 2, "B_paths": [""] * 2})
        model._forward_backward()
        model.sync_gradients()
        q.put((rank, [(o.flat_g * o.grad_scale).cpu().numpy() for o in model.optimizers],
               [o.flat_p.cpu().numpy() for o in model.optimizers]))
    finally:
        dist.destroy_process_group()


def test_global_mask_norm_equals_global_batch():
    """opt.global_mask_norm: two ranks x batch 2 give, after the gradient average, the gradient of ONE process on the
    global batch of 4 (the reference's DataParallel semantics, registration_model.py:160-166,262); without the option the
    per-rank 1/sum(mask) normalisation gives a measurably different gradient on shards with unequal masks."""
    A, B = _gmn_data()
    res = {}
    for gmn in (True, False):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_gmn, args=(r, 2, port, q, gmn)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            out = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
            for p in procs:
                p.join(timeout=120)
                assert p.exitcode == 0
        finally:
            _reap(procs)
        res[gmn] = out[0]
    # the single-process run on the global batch, from the ranks' (broadcast) weights
    model, opt = _gmn_model(4, "single")
    model.data_dependent_initialize({"A": A[:2].cuda().repeat(2, 1, 1, 1), "B": B[:2].cuda().repeat(2, 1, 1, 1),
                                     "A_paths": [""] * 4, "B_paths": [""] * 4})
    model.setup(opt)
    for o, w in zip(model.optimizers, res[True][2]):
        o.flat_p.copy_(torch.from_numpy(w))
    from dfmir_amd import ops
    ops.bump_weights_epoch()
    model.patch_id_source.call = 2          # as on the ranks: two init calls, then the step's three sets
    model.set_input({"A": A.cuda(), "B": B.cuda(), "A_paths": [""] * 4, "B_paths": [""] * 4})
    model._forward_backward()
    ref = [o.flat_g.cpu().numpy() for o in model.optimizers]

    def dist_to_ref(gs):
        return max(float(np.linalg.norm(g - r) / np.linalg.norm(r)) for g, r in zip(gs, ref))
    d_on, d_off = dist_to_ref(res[True][1]), dist_to_ref(res[False][1])
    assert d_on <= 2e-4, d_on
    assert d_off >= 20 * d_on and d_off >= 2e-3, (d_on, d_off)
