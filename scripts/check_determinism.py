"""GPU diagnostic: run the same training iterations twice and report, per parameter, where two runs differ (nothing
should: the library has no float atomics on the training path).  Usage: python scripts/check_determinism.py [c2|small]
[precision]"""
import os
import sys

os.environ.setdefault('GANGEALING_SYNTHETIC', '1')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch                                               # noqa: E402

from gangealing_amd.op import conv_mfma                    # noqa: E402
from gangealing_amd.train_step import GangealingTrainer    # noqa: E402

CFG = {'c2': dict(gen_size=256, flow_size=128, batch=16, transform=('similarity', 'flow'), inject=5, ndirs=1),
       'small': dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2),
       'cluster': dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2,
                       num_heads=2, flips=True, sample_from_full_res=True)}


def run(kw, steps):
    dev = torch.device('cuda', 0)
    tr = GangealingTrainer(dev, seed=5, stn_lr=1e-4, ll_lr=1e-4, perturb_heads=0.02, **kw)
    names = [n for n, _ in tr.stn.named_parameters()]
    recs = []
    for it in range(steps):
        torch.manual_seed(100 + it)
        parts = tr.step(psi=0.5)
        tr.flush()
        recs.append(dict(loss=parts['p'].clone(), tv=parts['tv'], grads=[p.grad.clone() for p in tr.stn.parameters()],
                         param=tr.stn_arena.param.clone(), ll_grad=tr.ll_arena.grad.clone()))
    torch.cuda.synchronize()
    return names, recs


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'small'
    prec = sys.argv[2] if len(sys.argv) > 2 else 'fp16x3'
    conv_mfma.set_precision(prec)
    names, a = run(CFG[cfg], 2)
    _, b = run(CFG[cfg], 2)
    bad = 0
    for it, (ra, rb) in enumerate(zip(a, b)):
        if not torch.equal(ra['loss'], rb['loss']):
            print(f'[{cfg} {prec}] it {it}: loss {float(ra["loss"])!r} vs {float(rb["loss"])!r}')
            bad += 1
        if ra['tv'] is not None and not torch.equal(ra['tv'], rb['tv']):
            print(f'[{cfg} {prec}] it {it}: tv differs')
            bad += 1
        if not torch.equal(ra['ll_grad'], rb['ll_grad']):
            print(f'[{cfg} {prec}] it {it}: latent-learner gradient differs')
            bad += 1
        for n, ga, gb in zip(names, ra['grads'], rb['grads']):
            if not torch.equal(ga, gb):
                d = (ga.double() - gb.double()).abs()
                print(f'[{cfg} {prec}] it {it}: grad {n} {tuple(ga.shape)}: {int((d > 0).sum())}/{d.numel()} differ, '
                      f'max {float(d.max()):.3e} (scale {float(ga.abs().max()):.3e})')
                bad += 1
        if not torch.equal(ra['param'], rb['param']):
            print(f'[{cfg} {prec}] it {it}: parameters differ after the update')
            bad += 1
    print(f'[{cfg} {prec}] ' + ('BITWISE IDENTICAL' if not bad else f'{bad} differences'))


if __name__ == '__main__':
    main()
