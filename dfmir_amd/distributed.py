"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" in CPU tests).  Replaces the reference's nn.DataParallel wrap
(models/base_model.py:103-107): weights are broadcast once, and after backward each network's flat
gradient arena is all-reduced as ONE collective (G 45.5 MB, F 2.2 MB, R 1.4 MB at 256^2) and
averaged by the fused Adam kernel's grad_scale.  No collective sits on the forward path."""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def broadcast_arena(flat_p, src=0):
    if is_distributed():
        dist.broadcast(flat_p, src=src)


def allreduce_arenas(flats, async_op=False):
    """Sum the flat gradient arenas over ranks (the 1/world average is applied in the Adam kernel)."""
    if not is_distributed():
        return []
    works = [dist.all_reduce(f, op=dist.ReduceOp.SUM, async_op=async_op) for f in flats]
    return works if async_op else []


def allreduce_scalars(values):
    """Average python floats / 0-dim tensors across ranks (logging only)."""
    if not is_distributed():
        return values
    t = torch.stack([v.detach().float().reshape(()) if torch.is_tensor(v) else torch.tensor(float(v)) for v in values])
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    t /= dist.get_world_size()
    return [float(x) for x in t.cpu()]
