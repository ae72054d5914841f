"""The N>1 trainer path on the hardware a test box has: two ranks share cuda:0 and exchange over gloo (RCCL refuses
two ranks on one device; the collective call sites are the same ones `backend='nccl'` runs on a multi-GPU node).
Each rank runs GangealingTrainer with pipeline_update=True - asynchronous all-reduce of the flat gradient arena,
Adam + EMA + re-pack deferred to the next STN forward - for three iterations and checks

  (i)   the replicas hold bit-identical STN / EMA / latent-learner parameters after every flush();
  (ii)  the gradient the optimiser consumed is exactly the sum of the two ranks' local gradients (the 1/world factor
        is folded into the Adam kernel), and the ranks really computed different local gradients;
  (iii) the run agrees BIT FOR BIT with the same three iterations in the immediate-update order
        (pipeline_update=False): parameters, EMA, latent learner, consumed gradients, losses.
"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu

KW = dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2, perturb_heads=0.02,
          seed=11)
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(trainer, dist, world, record):
    """STEPS iterations; after each flush() append CPU copies of what the checks need to `record`."""
    rank = dist.get_rank()
    consumed = []
    apply_update = trainer._apply_stn_update

    def spy_update(scale, lr):
        consumed.append(trainer.stn_arena.grad.clone())          # what Adam is about to read (after work.wait())
        return apply_update(scale, lr)
    trainer._apply_stn_update = spy_update
    real_all_reduce = dist.all_reduce
    local = []

    def spy_all_reduce(tensor, *args, **kw):
        if tensor.data_ptr() == trainer.stn_arena.grad.data_ptr():
            local.append(tensor.clone())                          # this rank's own gradient, before the exchange
        return real_all_reduce(tensor, *args, **kw)
    dist.all_reduce = spy_all_reduce
    try:
        for step in range(STEPS):
            torch.manual_seed(1000 * (rank + 1) + step)           # per-rank data stream
            parts = trainer.step(psi=0.5)
            trainer.flush()
            record.append(dict(param=trainer.stn_arena.param.cpu(), ema=trainer.ema_arena.param.cpu(),
                               ll=trainer.ll_arena.param.cpu(), consumed=consumed[-1].cpu(),
                               local=local[-1].cpu() if local else None, loss=float(parts['p'])))
    finally:
        dist.all_reduce = real_all_reduce
        trainer._apply_stn_update = apply_update


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from gangealing_amd import distributed as gdist
    from gangealing_amd.train_step import GangealingTrainer
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    assert gdist.setup_distributed('gloo') is True and gdist.get_world_size() == world
    result = dict(rank=rank)
    try:
        piped, immediate = [], []
        tr = GangealingTrainer(dev, pipeline_update=True, **KW)
        assert tr.world == world and tr.pipeline_update
        init = tr.stn_arena.param.cpu()
        _run(tr, dist, world, piped)
        tr2 = GangealingTrainer(dev, pipeline_update=False, **KW)
        _run(tr2, dist, world, immediate)
        ok_sync = ok_sum = True
        distinct = False
        for rec in piped:
            for key in ('param', 'ema', 'll'):
                both = [torch.empty_like(rec[key]) for _ in range(world)]
                dist.all_gather(both, rec[key])
                ok_sync = ok_sync and torch.equal(both[0], both[1])
            locs = [torch.empty_like(rec['local']) for _ in range(world)]
            dist.all_gather(locs, rec['local'])
            ok_sum = ok_sum and torch.equal(rec['consumed'], locs[0] + locs[1])
            distinct = distinct or not torch.equal(locs[0], locs[1])
        # (iii): same iterations, immediate order.  The library adds every partial sum in a fixed order (no float
        # atomics), the pipelined order shows every forward exactly the parameters of the immediate order, and gloo's
        # two-rank sum is commutative: the two runs must agree bit for bit.
        same_order = all(torch.equal(a[key], b[key]) for a, b in zip(piped, immediate)
                         for key in ('param', 'ema', 'll', 'consumed'))
        worst = max(float((a['param'].double() - b['param'].double()).abs().max()) for a, b in zip(piped, immediate))
        loss_same = all(a['loss'] == b['loss'] for a, b in zip(piped, immediate))
        result.update(ok_sync=bool(ok_sync), ok_sum=bool(ok_sum), distinct=bool(distinct), same_order=bool(same_order),
                      worst=worst, loss_same=bool(loss_same))
    except Exception as e:          # surface the failure in the parent instead of a queue timeout
        import traceback
        result['error'] = ''.join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:]
    q.put(result)
    gdist.synchronize()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_pipelined_trainer(cuda):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for res in results:
        assert 'error' not in res, res['error']
        assert res['ok_sync'], 'replicas diverged after flush()'
        assert res['ok_sum'] and res['distinct'], (res['ok_sum'], res['distinct'])
        assert res['same_order'] and res['loss_same'], ('pipelined and immediate update orders differ', res['worst'])
    for p in procs:
        assert p.exitcode == 0


def _withdrawn_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY='0')
    os.environ.pop('GG_ENABLE', None)
    import torch.distributed as dist
    from gangealing_amd import distributed as gdist
    from gangealing_amd.train_step import GangealingTrainer
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    assert gdist.setup_distributed('gloo') is True
    result = dict(rank=rank)
    try:
        try:
            GangealingTrainer(dev, use_graph=True, **KW)
            result['raised'] = False
        except RuntimeError as e:
            result['raised'] = 'nccl' in str(e) and 'gloo' in str(e)
        tr = GangealingTrainer(dev, use_graph='auto', **KW)        # batch 2 <= 8: 'auto' would pick replay in one process
        result['auto_is_eager'] = tr.collectives and not tr.use_graph
        parts = tr.step(psi=0.5)
        tr.flush()
        result['finite'] = bool(torch.isfinite(parts['p']))
    except Exception as e:
        import traceback
        result['error'] = ''.join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:]
    q.put(result)
    gdist.synchronize()
    dist.destroy_process_group()


def test_two_ranks_graph_replay_needs_a_capturable_backend(cuda):
    """use_graph=True with collectives captures the all-reduces inside the hipGraph, which only the nccl (RCCL) backend
    allows: on gloo it raises and names the reason; use_graph='auto' takes the eager pipelined step, which trains.  (The
    captured form itself runs in tests/test_gpu_rccl_single_rank.py on a one-rank RCCL group.)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_withdrawn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for res in results:
        assert 'error' not in res, res['error']
        assert res['raised'] is True and res['auto_is_eager'] and res['finite']
