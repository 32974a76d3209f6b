"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on
ROCm).  Replaces the reference's nn.DataParallel wrap (models/base_model.py:103-107): weights are
broadcast once, and after backward each network's flat gradient arena is all-reduced as ONE collective
(G 45.5 MB, F 2.2 MB, R 1.4 MB at 256^2) and averaged by the fused Adam kernel's grad_scale.  No
collective sits on the forward path.

`gloo` serves the tests: on CPU tensors as is, and on device tensors staged through the host -- two ranks
can then share ONE GPU (RCCL refuses two ranks per device), which is how the real model's multi-process
path is exercised on a 1-GPU box (tests/test_gpu_distributed.py).  DFMIR_DIST_BACKEND overrides the backend
chosen by `init_from_env` (bench.py, dfmir_amd.train)."""
import os

import torch
import torch.distributed as dist


def init_from_env(default_backend="nccl"):
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).  Returns
    (rank, world, device index).  With more ranks than visible GPUs (tests) ranks share devices round-robin."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(index)
    if (world > 1 or _FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DFMIR_DIST_BACKEND", default_backend)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, index


# DFMIR_FORCE_DIST=1: treat a world of ONE rank as distributed (process group created, weights broadcast, every arena
# all-reduced through the backend).  On a 1-GPU box this is the only way to run the RCCL code path itself -- communicator
# creation with device_id, the watchdog thread next to a hipGraph capture, async work handles -- before an 8-GPU node does.
_FORCE = os.environ.get("DFMIR_FORCE_DIST") not in (None, "", "0")     # flag semantics of include/dfmir_hip.h "Options"


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def _staged(t):
    """gloo + a device tensor: the collective runs on a host copy."""
    return t.is_cuda and dist.get_backend() == "gloo"


def broadcast_arena(flat_p, src=0):
    if not is_distributed():
        return
    if _staged(flat_p):
        h = flat_p.cpu()
        dist.broadcast(h, src=src)
        flat_p.copy_(h)
    else:
        dist.broadcast(flat_p, src=src)


def allreduce_arenas(flats, async_op=False):
    """Sum the flat gradient arenas over ranks (the 1/world average is applied in the Adam kernel)."""
    if not is_distributed():
        return []
    works = []
    for f in flats:
        if _staged(f):
            h = f.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            f.copy_(h)
            works.append(None)
        else:
            w = dist.all_reduce(f, op=dist.ReduceOp.SUM, async_op=async_op)
            works.append(w if async_op else None)
    return works if async_op else []


def allreduce_max(value, device):
    """max over ranks of a python float (bench timing)."""
    if not is_distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_float(value, device):
    """every rank's python float, in rank order (bench: per-rank step times)."""
    if not is_distributed():
        return [float(value)]
    dev = "cpu" if dist.get_backend() == "gloo" else device
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def barrier():
    if is_distributed():
        dist.barrier()


def allreduce_scalars(values):
    """Average python floats / 0-dim tensors across ranks (logging only)."""
    if not is_distributed():
        return values
    t = torch.stack([v.detach().float().reshape(()).cpu() if torch.is_tensor(v) else torch.tensor(float(v)) for v in values])
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    t /= dist.get_world_size()
    return [float(x) for x in t.cpu()]
