"""torch.distributed helpers with the names of utils/distributed.py:6-162.  One process per GPU;
backend "nccl" is RCCL on ROCm (xGMI inside a node).  Every helper degrades to a no-op at
world_size 1; the backend can be overridden (gloo) for CPU tests."""
import os

import torch
from torch import distributed as dist


def setup_distributed(backend=None):
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC (the host driver's only supported mode)
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend, init_method='env://')
        synchronize()
    return world > 1


def _on():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if _on() else 0


def get_world_size():
    return dist.get_world_size() if _on() else 1


def primary():
    return get_rank() == 0


def synchronize():
    if _on() and dist.get_world_size() > 1:
        dist.barrier()


def all_gather(input, cat=True):
    if get_world_size() == 1:
        return input if cat else input.unsqueeze(0)
    out = [torch.empty_like(input) for _ in range(get_world_size())]
    dist.all_gather(out, input.contiguous())
    return torch.cat(out, 0) if cat else torch.stack(out, 0)


def all_reduce_mean_(flat):
    """In-place mean all-reduce of one flat buffer (the whole STN gradient arena in a single
    collective: ring all-reduce over xGMI is per-link bound, so one large message beats 25 MB buckets)."""
    if get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(get_world_size())
    return flat


def rank0_to_all(input):
    if get_world_size() > 1:
        dist.broadcast(input, src=0)
    return input


def reduce_loss_dict(loss_dict):
    """Mean of every loss over ranks, valid on rank 0 (utils/distributed.py:140-162)."""
    world = get_world_size()
    if world < 2:
        return loss_dict
    with torch.no_grad():
        keys = sorted(loss_dict.keys())
        losses = torch.stack([loss_dict[k].detach().float().reshape(()) for k in keys], 0)
        dist.reduce(losses, dst=0)
        if dist.get_rank() == 0:
            losses /= world
        return {k: v for k, v in zip(keys, losses)}
