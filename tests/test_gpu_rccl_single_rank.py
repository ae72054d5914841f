"""RCCL on the hardware a test box has: a ONE-rank `nccl` process group (backend "nccl" IS RCCL on ROCm).

The 1-GPU boxes cannot run the 2/4/8-rank job of train.py:256-259 / utils/distributed.py:6-14, and RCCL refuses two
ranks on one device - but a one-rank group still loads librccl, creates a communicator, launches its all-reduce kernel
on RCCL's own stream and hands the result back to the compute stream through `work.wait()`.  That is every moving part
of the multi-GPU step except the xGMI rings.  GangealingTrainer(collectives=True) forces the `world > 1` call sequence
of train_step.py (step(): small all-reduce of the latent gradient, ASYNC all-reduce of the real 172 MB STN gradient
arena; flush(): work.wait(), Adam + EMA + weight re-pack behind the next iteration's generator passes) on that group.

Checked: the backend really is nccl and counts one rank; the arena pushed through the collective is the full-size one
(C2's STN: 43 M parameters); three iterations agree BIT FOR BIT with the same iterations without any collective (a
one-rank sum is the identity, so any difference is a stream hand-off bug: Adam reading the arena before the collective
wrote it back, or the next backward writing gradients into a buffer RCCL is still reading); the exposed wait is
recorded.

Round 6: the hipGraph form of the multi-process step - `GangealingTrainer(collectives=True, use_graph=True)` captures both
all-reduces INSIDE the one graph of the iteration (RCCL's kernels on the communicator's stream, event-forked from the
capturing stream).  On the one-rank group: the capture succeeds, five replays run, and parameters / EMA / latent learner /
losses agree bit for bit with the eager collective step and with the step that has no collective at all.
"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu

KW = dict(gen_size=256, flow_size=128, batch=2, transform=('similarity', 'flow'), inject=5, ndirs=1, perturb_heads=0.02,
          seed=5, stn_lr=1e-4, ll_lr=1e-4)
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _iterate(trainer):
    out = []
    for step in range(STEPS):
        torch.manual_seed(100 + step)
        parts = trainer.step(psi=0.5)
        trainer.flush()
        out.append(dict(param=trainer.stn_arena.param.clone(), ema=trainer.ema_arena.param.clone(),
                        ll=trainer.ll_arena.param.clone(), loss=float(parts['p'])))
    return out


def _worker(port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0', GANGEALING_SYNTHETIC='1')
    result = {}
    try:
        import torch.distributed as dist
        from gangealing_amd.train_step import GangealingTrainer
        torch.cuda.set_device(0)
        dev = torch.device('cuda', 0)
        dist.init_process_group(backend='nccl', init_method='env://', world_size=1, rank=0)
        result['backend'] = dist.get_backend()
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        result['ranks_counted'] = float(ones.item())

        calls = []
        real = dist.all_reduce

        def spy(tensor, *a, **kw):
            calls.append((tensor.numel(), bool(kw.get('async_op', False))))
            return real(tensor, *a, **kw)
        dist.all_reduce = spy
        tr = GangealingTrainer(dev, collectives=True, **KW)
        assert tr.world == 1 and tr.collectives and tr.pipeline_update
        tr.comm_events = []
        piped = _iterate(tr)
        torch.cuda.synchronize()
        dist.all_reduce = real
        result['arena_numel'] = int(tr.stn_arena.numel)
        result['calls'] = list(calls)
        result['exposed_ms'] = [a.elapsed_time(b) for a, b in tr.comm_events]
        del tr
        plain = _iterate(GangealingTrainer(dev, collectives=False, **KW))
        result['bitwise'] = all(torch.equal(a[k], b[k]) for a, b in zip(piped, plain) for k in ('param', 'ema', 'll'))
        result['loss_same'] = all(a['loss'] == b['loss'] for a, b in zip(piped, plain))
        result['worst'] = max(float((a['param'].double() - b['param'].double()).abs().max())
                              for a, b in zip(piped, plain))
        result['moved'] = float((piped[-1]['param'] - piped[0]['param']).abs().max())
        # ---- the same iterations with the all-reduces captured inside the iteration's hipGraph
        try:
            trg = GangealingTrainer(dev, collectives=True, use_graph=True, graph_warmup=1, **KW)
            assert trg.collectives and trg.use_graph and not trg.pipeline_update
            calls.clear()
            dist.all_reduce = spy
            graphed = _iterate(trg)                  # call 1 eager, call 2 captures + replays, call 3 replays
            more = []
            for step in range(STEPS, STEPS + 2):     # two more replays
                torch.manual_seed(100 + step)
                more.append(float(trg.step(psi=0.5)['p']))
            torch.cuda.synchronize()
            dist.all_reduce = real
            result['graph_captured'] = trg._graph is not None
            result['graph_steps'] = (trg.stn_arena.step_count, trg.ll_arena.step_count)
            result['graph_params_finite'] = bool(torch.isfinite(trg.stn_arena.param).all())
            # python-level all_reduce calls: 2 in the eager warm-up iteration + 2 while capturing; replays issue none
            result['graph_allreduce_calls'] = len(calls)
            del trg
            # reference: the same graph path with NO collective in it (a one-rank sum is the identity, so any difference
            # is a hand-off bug between the capturing stream and RCCL's stream inside the graph) ...
            solo = _iterate(GangealingTrainer(dev, collectives=False, use_graph=True, graph_warmup=1, **KW))
            result['graph_bitwise'] = all(torch.equal(a[k], b[k]) for a, b in zip(graphed, solo)
                                          for k in ('param', 'ema', 'll'))
            result['graph_loss_same'] = all(a['loss'] == b['loss'] for a, b in zip(graphed, solo))
            # ... and the eager step (recorded; see the assertion)
            result['graph_vs_eager'] = max(float((a['param'].double() - b['param'].double()).abs().max())
                                           for a, b in zip(graphed, plain))
            result['graph_more_finite'] = all(l == l and abs(l) != float('inf') for l in more)
        except Exception as e:
            import traceback
            result['graph_error'] = ''.join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:]
        dist.destroy_process_group()
    except Exception as e:          # surface the failure in the parent instead of a queue timeout
        import traceback
        result['error'] = ''.join(traceback.format_exception(type(e), e, e.__traceback__))[-3000:]
    q.put(result)


def test_single_rank_rccl_all_reduce_of_the_real_arena(cuda):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=900)
    p.join(timeout=120)
    assert 'error' not in res, res['error']
    assert res['backend'] == 'nccl' and res['ranks_counted'] == 1.0
    # the full-size STN arena (C2: ~43 M fp32 = 172 MB) went through RCCL asynchronously, once per iteration
    assert res['arena_numel'] > 40e6
    big = [c for c in res['calls'] if c[0] == res['arena_numel']]
    assert len(big) == STEPS and all(is_async for _, is_async in big), res['calls']
    assert len(res['exposed_ms']) == STEPS
    assert res['moved'] > 0
    assert res['bitwise'] and res['loss_same'], ('collective path differs from the plain path', res['worst'])
    # the all-reduces captured inside the iteration's hipGraph (round 6)
    assert 'graph_error' not in res, res['graph_error']
    assert res['graph_captured'] and res['graph_allreduce_calls'] == 4, res['graph_allreduce_calls']
    assert res['graph_bitwise'] and res['graph_loss_same'], 'captured collectives changed the replayed iteration'
    # against the eager step only a sanity bound: a replayed graph draws its latents / noise from the graph-registered
    # Philox state, not from the eager generator's stream - other samples, so Adam's first steps (+-lr per entry) differ
    assert res['graph_vs_eager'] <= 2 * STEPS * KW['stn_lr'] * 1.01, res['graph_vs_eager']
    assert res['graph_more_finite'] and res['graph_params_finite'] and res['graph_steps'] == (STEPS + 2, STEPS + 2)
    assert p.exitcode == 0
