"""Run-to-run reproducibility of the product itself.  The library has no floating-point atomics on the training path
(include/gangealing_hip.h, "reproducibility"): split-K convolutions, K-split weight gradients, grid-wide sums and the
scatter-shaped gradients all add their partial results in a fixed order.  Two executions of the same iterations from the
same seeds must therefore agree BIT FOR BIT - losses, every gradient, parameters, Adam moments, the EMA copy."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIGS = {
    # BASELINE.json configs[1] at its real shapes and batch: the tile variants, split-K layers (<= 16^2), row-streaming
    # and generic weight gradients the benchmark runs
    'c2-batch16': dict(gen_size=256, flow_size=128, batch=16, transform=('similarity', 'flow'), inject=5, ndirs=1,
                       perturb_heads=0.02),
    # small: the batch sizes at which rounds 2 - 4 forked the generator passes onto two streams (opt-in since round 5:
    # losses.sample_gan_supervised_pairs), similarity-only and clustering heads
    'small-flow': dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2,
                       perturb_heads=0.02),
    'small-cluster': dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2,
                          num_heads=2, flips=True, perturb_heads=0.02, sample_from_full_res=True),
}


def run(cuda, kw, steps, precision):
    from gangealing_amd.op import conv_mfma
    from gangealing_amd.train_step import GangealingTrainer
    old = conv_mfma.PRECISION
    conv_mfma.set_precision(precision)
    try:
        tr = GangealingTrainer(cuda, seed=5, stn_lr=1e-4, ll_lr=1e-4, **kw)
        out = []
        for it in range(steps):
            torch.manual_seed(100 + it)
            parts = tr.step(psi=0.5)
            tr.flush()
            out.append(dict(p=parts['p'].clone(), tv=None if parts['tv'] is None else parts['tv'].clone(),
                            grad=tr.stn_arena.grad.clone(), ll_grad=tr.ll_arena.grad.clone(),
                            param=tr.stn_arena.param.clone(), ema=tr.ema_arena.param.clone(),
                            m=tr.stn_arena.exp_avg.clone(), v=tr.stn_arena.exp_avg_sq.clone(),
                            ll=tr.ll_arena.param.clone()))
        torch.cuda.synchronize()
        return out
    finally:
        conv_mfma.set_precision(old)


@pytest.mark.parametrize('precision', ['bf16x3', 'fp16x3', 'fp32'])
@pytest.mark.parametrize('name', sorted(CONFIGS))
def test_train_iterations_are_bitwise_reproducible(name, precision, cuda):
    steps = 2
    a = run(cuda, CONFIGS[name], steps, precision)
    b = run(cuda, CONFIGS[name], steps, precision)
    for it, (ra, rb) in enumerate(zip(a, b)):
        for key in ra:
            if ra[key] is None:
                continue
            same = torch.equal(ra[key], rb[key])
            if not same:
                d = (ra[key].double() - rb[key].double()).abs()
                pytest.fail(f'{name} [{precision}] iteration {it}: `{key}` differs between two runs '
                            f'({int((d > 0).sum())} of {d.numel()} entries, max {float(d.max()):.3e})')
    assert float((a[-1]['param'] - a[0]['param']).abs().max()) > 0          # the iterations really trained
