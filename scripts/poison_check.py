"""Does any kernel of the training step read memory it (or a predecessor) never wrote?

    python scripts/poison_check.py [config ...]          (configs: small, cluster, res128; default: all)

Every buffer the package takes UNINITIALISED from torch (torch.empty / empty_like / new_empty / empty_strided, including
the library's scratch through gangealing_amd/_lib.py's allocator callback) is filled with a poison pattern first: NaN for
floating types, all-ones for integers (0xFFFF is a NaN in both 16-bit limb formats).  The library is bitwise reproducible,
so three seeded iterations with the poison must equal three without it bit for bit - a read of an unwritten element, even
one that is multiplied by zero, turns into a NaN or a different bit somewhere downstream.  Prints the first module whose
output is not finite and every parameter whose gradient differs."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault('GANGEALING_SYNTHETIC', '1')

CONFIGS = {
    'small': dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2),
    'cluster': dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2, num_heads=2,
                    flips=True, sample_from_full_res=True),
    'res128': dict(gen_size=128, flow_size=64, batch=3, transform=('similarity', 'flow'), inject=5, ndirs=4),
}

_ORIG = {}
_STATE = {'on': False, 'count': 0}


def _poison(t):
    if _STATE['on'] and isinstance(t, torch.Tensor) and t.is_cuda and t.numel() and not torch.cuda.is_current_stream_capturing():
        _STATE['count'] += 1
        if t.dtype.is_floating_point:
            t.fill_(float('nan'))
        elif t.dtype == torch.bool:
            t.fill_(True)
        elif t.dtype == torch.uint8:
            t.fill_(255)
        else:
            t.fill_(-1)
    return t


def install():
    for name in ('empty', 'empty_like', 'empty_strided'):
        _ORIG[name] = getattr(torch, name)
        setattr(torch, name, (lambda f: lambda *a, **k: _poison(f(*a, **k)))(_ORIG[name]))
    _ORIG['new_empty'] = torch.Tensor.new_empty
    torch.Tensor.new_empty = lambda self, *a, **k: _poison(_ORIG['new_empty'](self, *a, **k))


def run(cfg, poisoned, steps=3):
    from gangealing_amd.train_step import GangealingTrainer
    dev = torch.device('cuda', 0)
    _STATE['on'] = poisoned
    _STATE['count'] = 0
    tr = GangealingTrainer(dev, perturb_heads=0.02, seed=5, stn_lr=1e-4, ll_lr=1e-4, **CONFIGS[cfg])
    first_bad = []
    if poisoned:
        def hook(name):
            def f(mod, inp, out):
                outs = out if isinstance(out, (tuple, list)) else (out,)
                for o in outs:
                    if isinstance(o, torch.Tensor) and o.dtype.is_floating_point and not first_bad:
                        if not bool(torch.isfinite(o).all()):
                            first_bad.append(name + ' (' + type(mod).__name__ + ')')
            return f
        for root, net in (('generator', tr.generator), ('stn', tr.stn), ('ll', tr.ll), ('loss_fn', tr.loss_fn)):
            if isinstance(net, torch.nn.Module):
                for name, mod in net.named_modules():
                    mod.register_forward_hook(hook(root + '.' + name))
    rec = []
    for step in range(steps):
        torch.manual_seed(100 + step)
        parts = tr.step(psi=0.5)
        tr.flush()
        torch.cuda.synchronize()
        rec.append(dict(loss={k: v.clone() for k, v in parts.items() if v is not None},
                        stn_grad=tr.stn_arena.grad.clone(), ll_grad=tr.ll_arena.grad.clone(),
                        grads=[p.grad.clone() for p in tr.stn.parameters()],
                        stn_param=tr.stn_arena.param.clone(), ll_param=tr.ll_arena.param.clone(),
                        ema=tr.ema_arena.param.clone()))
    _STATE['on'] = False
    names = [n for n, _ in tr.stn.named_parameters()]
    return rec, first_bad, _STATE['count'], tr, names


def same(a, b):
    return a.shape == b.shape and bool(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a,
                                                  b.view(torch.int32) if b.dtype == torch.float32 else b))


def main():
    configs = sys.argv[1:] or list(CONFIGS)
    install()
    bad = 0
    for cfg in configs:
        for prec in ('fp16x3', 'fp32'):
            from gangealing_amd.op import conv_mfma
            conv_mfma.set_precision(prec)
            got, first_bad, count, tr, names = run(cfg, True)     # (first: the library's scratch grows poisoned)
            del tr
            ref, _, _, tr, _ = run(cfg, False)
            del tr
            diffs = []
            for step, (r, g) in enumerate(zip(ref, got)):
                for key in r:
                    if key == 'loss':
                        for k in r['loss']:
                            if not same(r['loss'][k], g['loss'][k]):
                                diffs.append(f'step {step} loss.{k}: {float(r["loss"][k]):.6g} vs {float(g["loss"][k]):.6g}')
                    elif key == 'grads':
                        for n, a, b in zip(names, r['grads'], g['grads']):
                            if not same(a, b) and len(diffs) < 40:
                                diffs.append(f'step {step} grad {n} {tuple(a.shape)}: {int(torch.isnan(b).sum())} NaN, '
                                             f'{int((a != b).sum())} of {a.numel()} differ')
                    elif not same(r[key], g[key]):
                        nan = int(torch.isnan(g[key]).sum())
                        diffs.append(f'step {step} {key}: {nan} NaN, {int((r[key] != g[key]).sum())} of {r[key].numel()} differ')
            status = 'IDENTICAL' if not diffs else 'DIFFERENT'
            print(f'[{cfg} {prec}] {count} poisoned buffers: poisoned run {status} to the clean run', flush=True)
            if first_bad:
                print(f'    first module with a non-finite output: {first_bad[0]}', flush=True)
            for line in diffs[:40]:
                print('    ' + line, flush=True)
            bad += bool(diffs)
            torch.cuda.empty_cache()
    print('POISON CHECK ' + ('PASSED' if bad == 0 else f'FAILED for {bad} runs'), flush=True)
    return 0 if bad == 0 else 1


if __name__ == '__main__':
    sys.exit(main())
