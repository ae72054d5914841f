"""Round 5, session 9 (debug): where do the two replicas of the segment-replay trainer part?"""
import os, socket, sys
import torch, torch.multiprocessing as mp
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KW = dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2, perturb_heads=0.02, seed=11)


def worker(rank, world, port, flushed):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY='0', GANGEALING_SYNTHETIC='1')
    import torch.distributed as dist
    from gangealing_amd import distributed as gdist
    from gangealing_amd.train_step import GangealingTrainer
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    gdist.setup_distributed('gloo')
    tr = GangealingTrainer(dev, use_graph=True, graph_warmup=2, **KW)
    for step in range(8):
        torch.manual_seed(1000 * (rank + 1) + step)
        tr.step(psi=0.5)
        if flushed:
            tr.flush()
        torch.cuda.synchronize()
        out = []
        for name, arena in (('stn', tr.stn_arena), ('ema', tr.ema_arena), ('ll', tr.ll_arena)):
            for what in ('param', 'grad'):
                t = getattr(arena, what, None)
                if t is None:
                    continue
                mine = t.detach().cpu()
                both = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(both, mine)
                d = float((both[0].double() - both[1].double()).abs().max())
                out.append(f'{name}.{what} {d:.2e}/{float(both[0].abs().max()):.1e}')
        if rank == 0:
            print(f'flushed={flushed} step {step} segments={tr._segments is not None} pending={tr._pending_work is not None}: ' + '  '.join(out), flush=True)
    tr.flush()
    dist.destroy_process_group()


if __name__ == '__main__':
    for flushed in ((True,) if 'flushed' in sys.argv[1:] else (True, False)):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context('spawn')
        ps = [ctx.Process(target=worker, args=(r, 2, port, flushed)) for r in range(2)]
        [p.start() for p in ps]
        [p.join(600) for p in ps]
