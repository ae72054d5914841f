"""CPU suite: the N>1 path (one process per device, single flat-arena all-reduce, per-rank data
seeds, rank-0 loss reduction) exercised with world_size 2 on the gloo backend."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gangealing_amd import distributed as gdist
    from gangealing_amd.train_step import FlatArena
    assert gdist.setup_distributed('gloo') is True
    assert gdist.get_world_size() == world and gdist.get_rank() == rank and gdist.primary() == (rank == 0)
    torch.manual_seed(0)                                   # identical replicas (as GangealingTrainer does)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 2))
    arena = FlatArena(net)
    torch.manual_seed(0 * world + rank)                    # per-rank data stream (train.py:193-194)
    x = torch.randn(4, 6)
    arena.zero_grad()
    loss = net(x).pow(2).mean()
    loss.backward()
    local = arena.grad.clone()
    gdist.all_reduce_mean_(arena.grad)                     # ONE collective over the whole arena
    gathered = gdist.all_gather(local.unsqueeze(0))
    expect = gathered.mean(dim=0)
    ok_grad = torch.allclose(arena.grad, expect, atol=1e-7)
    # plain SGD step on the averaged gradient keeps replicas identical
    with torch.no_grad():
        arena.param.add_(arena.grad, alpha=-0.1)
    params = gdist.all_gather(arena.param.unsqueeze(0))
    ok_sync = torch.equal(params[0], params[1])
    red = gdist.reduce_loss_dict({'p': loss, 'tv': loss * 2})
    all_losses = gdist.all_gather(loss.detach().reshape(1))
    ok_red = True
    if rank == 0:
        ok_red = torch.allclose(red['p'], all_losses.mean()) and torch.allclose(red['tv'], 2 * all_losses.mean())
    distinct = not torch.equal(gathered[0], gathered[1])    # ranks really saw different data
    b = torch.tensor([float(rank + 5)])
    gdist.rank0_to_all(b)
    q.put((rank, bool(ok_grad), bool(ok_sync), bool(ok_red), bool(distinct), float(b)))
    gdist.synchronize()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_grad, ok_sync, ok_red, distinct, b in results:
        assert ok_grad and ok_sync and ok_red and distinct and b == 5.0, (rank, ok_grad, ok_sync, ok_red, distinct, b)


def test_helpers_degrade_at_world_size_1():
    from gangealing_amd import distributed as gdist
    assert gdist.get_world_size() == 1 and gdist.get_rank() == 0 and gdist.primary()
    t = torch.arange(3.0)
    assert torch.equal(gdist.all_gather(t), t) and torch.equal(gdist.all_reduce_mean_(t.clone()), t)
    assert gdist.reduce_loss_dict({'a': t})['a'] is t
    gdist.synchronize()
