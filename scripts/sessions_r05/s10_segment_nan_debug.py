"""Round 5, session 10 (debug): which tensor of the segment-replay step turns NaN at the second replay?

    python s10_segment_nan_debug.py <label> [single]

Two ranks on one GPU over gloo (or one process with `single`: the one-graph path).  GG_DISABLE / GG_ENABLE come from
the environment.  After every step rank 0 prints: the loss parts, whether the generator outputs / the STN output that the
captured graphs left behind are finite, whether every registered weight pack is finite and equal to a fresh pack of the
current weight, and whether the parameters are finite."""
import os, socket, sys
import torch, torch.multiprocessing as mp
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KW = dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2, perturb_heads=0.02, seed=11)


def fin(t):
    if t is None:
        return 'none'
    t = t.detach()
    if t.dtype == torch.int16:
        return 'int'
    ok = bool(torch.isfinite(t.float()).all())
    return f'ok({float(t.float().abs().max()):.2e})' if ok else 'NAN'


def pack_report(conv_mfma, _lib):
    reg = conv_mfma.TRAINABLE_PACKS
    if reg is None:
        return 'no registry'
    bad, stale, n = 0, 0, 0
    for key, (pw, ver) in list(reg.entries.items()):
        for limbs, (buf, cnt) in pw._split.items():
            n += 1
            fresh = torch.empty_like(buf)
            _lib.call('gg_conv_pack_weight_split', fresh, pw.weight.contiguous(), pw.groups, pw.cout_g, pw.cin_g, pw.k,
                      pw.k, pw.transpose_io, pw.flip, pw.scale, limbs)
            if not torch.equal(fresh, buf):
                stale += 1
            view = buf.view(torch.float16 if limbs & 16 else torch.bfloat16)
            if not bool(torch.isfinite(view.float()).all()):
                bad += 1
        if pw._fp32 is not None:
            n += 1
            if not bool(torch.isfinite(pw._fp32).all()):
                bad += 1
    return f'packs {n}: {bad} non-finite, {stale} differ from a fresh pack; jobs {reg.njobs}'


def worker(rank, world, port, label):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY='0', GANGEALING_SYNTHETIC='1')
    from gangealing_amd import distributed as gdist, losses, _lib
    from gangealing_amd.op import conv_mfma
    from gangealing_amd.train_step import GangealingTrainer
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    if world > 1:
        gdist.setup_distributed('gloo')
    stash = {}
    orig = losses.sample_gan_supervised_pairs

    def sampled(*a, **k):
        u, t = orig(*a, **k)
        stash['unaligned'], stash['target'] = u, t
        return u, t
    if os.environ.get('S10_NOSTASH') != '1':
        losses.sample_gan_supervised_pairs = sampled
    tr = GangealingTrainer(dev, use_graph=True, graph_warmup=2, **KW)
    if os.environ.get('S10_NOSTASH') != '1':
        tr.stn.register_forward_hook(lambda m, i, o: stash.__setitem__('stn_out', o[0] if isinstance(o, (tuple, list)) else o))
    for step in range(7):
        torch.manual_seed(1000 * (rank + 1) + step)
        parts = tr.step(psi=0.5)
        torch.cuda.synchronize()
        if rank == 0:
            seg = tr._segments is not None or getattr(tr, '_graph', None) is not None
            line = [f'[{label}] step {step} replay={seg}',
                    'loss ' + ' '.join(f'{k}={float(v):.4g}' for k, v in parts.items() if v is not None),
                    'unaligned ' + fin(stash.get('unaligned')), 'target ' + fin(stash.get('target')),
                    'stn_out ' + fin(stash.get('stn_out')),
                    'stn.grad ' + fin(tr.stn_arena.grad), 'll.grad ' + fin(tr.ll_arena.grad),
                    'stn.param ' + fin(tr.stn_arena.param), 'll.param ' + fin(tr.ll_arena.param),
                    'stn.m ' + fin(tr.stn_arena.exp_avg), 'stn.v ' + fin(tr.stn_arena.exp_avg_sq)]
            print('  '.join(line), flush=True)
            print(f'[{label}]     before the deferred update: ' + pack_report(conv_mfma, _lib), flush=True)
        tr.flush()
        torch.cuda.synchronize()
        if rank == 0:
            print(f'[{label}]     after the update: ' + pack_report(conv_mfma, _lib) + '  stn.param ' + fin(tr.stn_arena.param), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    label = sys.argv[1]
    world = 1 if 'single' in sys.argv[2:] else 2
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    if world == 1:
        worker(0, 1, port, label)
    else:
        ctx = mp.get_context('spawn')
        ps = [ctx.Process(target=worker, args=(r, 2, port, label)) for r in range(2)]
        [p.start() for p in ps]
        [p.join(300) for p in ps]
