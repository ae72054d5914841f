"""The LITERAL drop-in on hardware: the reference's own Python, unmodified, on the HIP operators.

`gangealing_amd.launch.inject(root)` is what `python -m gangealing_amd.launch train.py ...` does before it runs the
script: the reference's operator packages (models.stylegan2.op.{upfirdn2d,fused_act,conv2d_gradfix},
utils.splat2d_cuda, models.spatial_transformers.antialiased_sampling) are pre-populated in sys.modules with this
package's modules, and every other reference module - networks.py (ModulatedConv2d in its per-sample-weight /
groups = N form, networks.py:233-282), warping_heads.py, spatial_transformer.py, latent_learner.py, loss.py, lpips.py -
is imported from the reference as it is.  Here those modules are built on cuda:0 and driven by the SAME driver
(oracle/config_cases.run_config) that produced the cfg_* fixtures from the reference on its CPU fallback, so this
compares  reference Python + HIP operators on the MI355X  with  reference Python + reference CPU operators.

The reference's Python comes from oracle/_ref/pyref (staged by `make -C oracle` from the checkout; git-ignored, travels
with gpurun like oracle/_ref/libsplat_ref.so) or from the checkout itself where that exists.  Everything that is not one
of the drop-in operators (EqualLinear, affine_grid, unfold / softmax of the convex upsampling, the VGG trunk of the
reference's LPIPS class) runs on ATen / MIOpen, as it would for a user of the launcher.
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, record_parity
from test_gpu_configs import (ACT_TOL, SIM_FACTOR, SIM_FLOOR_SCALE, check_batch, check_grads)

pytestmark = pytest.mark.gpu

_API = None


def reference_api():
    global _API
    if _API is None:
        from oracle import pyref
        if pyref.find_root() is None:
            pytest.skip('reference Python not staged (run `make -C oracle` where /root/reference exists)')
        _API = pyref.hip_api()
    return _API


@pytest.fixture(params=['fp32', 'fp16x3'])
def mode(request):
    from gangealing_amd.op import conv_mfma
    old = conv_mfma.PRECISION
    conv_mfma.set_precision(request.param)
    yield request.param
    conv_mfma.set_precision(old)


def test_reference_modules_resolve_to_hip_operators(cuda):
    """What the reference's networks.py / warping_heads.py bound at import time ARE this package's operators."""
    api = reference_api()
    import models.stylegan2.networks as ref_networks
    import models.spatial_transformers.warping_heads as ref_heads
    import gangealing_amd.op as op
    from gangealing_amd.spatial_transformers import antialiased_sampling as aa
    assert api.Generator.__module__ == 'models.stylegan2.networks'
    assert os.path.samefile(os.path.dirname(os.path.dirname(os.path.dirname(ref_networks.__file__))), api.root)
    assert ref_networks.upfirdn2d is op.upfirdn2d and ref_networks.fused_leaky_relu is op.fused_leaky_relu
    assert ref_networks.FusedLeakyReLU is op.FusedLeakyReLU and ref_networks.conv2d_gradfix is op.conv2d_gradfix
    assert ref_heads.MipmapWarp is aa.MipmapWarp and ref_heads.Warp is aa.Warp
    # and the reference's generator, on the GPU, against the reference's generator on its CPU fallback
    (c,) = load_golden('c2_generator')
    from test_gpu_configs import load_det, D
    g = load_det(api.Generator(256, 512, 8)).to(cuda).eval().requires_grad_(False)
    n = c['meta']['batch']
    noise = [D(f'c2gen.noise{i}', (n, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)), cuda) for i in range(g.num_layers)]
    with torch.no_grad():
        img, _ = g([torch.from_numpy(c['z']).to(cuda)], return_latents=True, noise=noise)
    check_batch('reference_dropin/c2_generator', 'default', img, c, 'img')


@pytest.mark.parametrize('name', ['c1', 'c2'])
def test_reference_train_loss_step_on_hip_operators(name, mode, cuda):
    """train.py:106-124 evaluated by the reference's own modules on cuda:0 through the HIP operators, at BASELINE
    configs[0] (64^2, similarity STN, batch 4) and configs[1] (256^2, similarity+flow STN @128^2, batch 16: the
    generator's grouped convolutions run with groups = 16) - activations at 1e-4, every loss term at 1e-4 relative,
    gradients by the rule of tests/test_gpu_configs.py."""
    from oracle import config_cases as cc
    api = reference_api()
    (c,) = load_golden(f'cfg_{name}')
    res = cc.run_config(api, name, cuda)
    test = f'reference_dropin/cfg_{name}'
    for key in ('unaligned', 'target', 'pred', 'stn_delta', 'delta_flow'):
        assert list(res[key].shape) == c['meta']['shapes'][key], (key, res[key].shape)
        check_batch(test, mode, res[key], c, key, tol=ACT_TOL)
    has_flow = 'flow' in cc.CONFIGS[name]['transform']
    for key in ('ploss', 'total') + (('tv', 'identity') if has_flow else ()):
        err = record_parity(test, mode, key, res[key].cpu().numpy(), c[key])
        assert err <= 1e-4 * abs(float(c[key])) + 1e-9, (key, err, float(c[key]))
    grads = res['grads']
    assert set(grads) == set(c['meta']['grad_norms'])
    if has_flow:
        check_grads(test + '/flow-stage', mode, grads, c, select=lambda k: k.startswith('stns.1.'))
        check_grads(test + '/similarity-stage', mode, grads, c, select=lambda k: k.startswith('stns.0.'),
                    factor=SIM_FACTOR, floor_scale=SIM_FLOOR_SCALE)
    else:
        check_grads(test + '/similarity-stage', mode, grads, c, select=lambda k: k != 'll.coefficients',
                    factor=SIM_FACTOR, floor_scale=SIM_FLOOR_SCALE)
    check_grads(test + '/latent-learner', mode, grads, c, select=lambda k: k == 'll.coefficients')


def test_reference_training_iterations_run(cuda):
    """Three iterations of train.py:106-134 with the reference's modules and torch.optim.Adam on cuda:0: losses stay
    finite, the STN parameters move, the EMA copy follows (models/__init__.py:19-24)."""
    api = reference_api()
    torch.manual_seed(0)
    gen = api.Generator(64, 512, 8).to(cuda).eval().requires_grad_(False)
    stn = api.get_stn(['similarity', 'flow'], flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=1).to(cuda)
    ema = api.get_stn(['similarity', 'flow'], flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=1).to(cuda)
    api.accumulate(ema, stn, 0)
    ll = api.DirectionInterpolator(None, 1, 5, gen.n_latent, 1).to(cuda)
    net = api.LPIPS(net='vgg', lpips=False, pnet_rand=True, pretrained=False, verbose=False).to(cuda).eval()
    loss_fn = lambda x, y: net(x, y) / 18.0
    t_optim = torch.optim.Adam(stn.parameters(), lr=1e-4)
    ll_optim = torch.optim.Adam(ll.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in stn.parameters()]
    losses = []
    for _ in range(3):
        ploss, delta = api.gangealing_loss(gen, stn, ll, loss_fn, torch.nn.Sequential(), 0.5, 4, 512, False, cuda,
                                           padding_mode='reflection')
        total = ploss + 1000.0 * api.total_variation_loss(delta) + api.flow_identity_loss(delta)
        stn.zero_grad()
        ll.zero_grad()
        total.backward()
        t_optim.step()
        ll_optim.step()
        api.accumulate(ema, stn, 0.5 ** (32 / 10000))
        losses.append(float(total.detach()))
    assert all(np.isfinite(losses)), losses
    moved = sum(float((p.detach() - b).abs().sum()) for p, b in zip(stn.parameters(), before))
    assert moved > 0
    drift = sum(float((e - p).abs().sum()) for e, p in zip(ema.parameters(), stn.parameters()))
    assert np.isfinite(drift) and drift > 0


def test_launcher_modules_route_trains(cuda):
    """`python -m gangealing_amd.launch --modules train.py`: the reference's models/__init__.py, latent_learner.py and
    training glue with this package's generator / STN / loss modules bound under the reference's module names.  The names
    train.py imports resolve to this package's classes, three iterations of train.py:106-134 run, and the first
    iteration's loss equals the one this package's own modules give from the same seed (they ARE the same modules)."""
    from oracle import pyref
    if pyref.find_root() is None:
        pytest.skip('reference Python not staged (run `make -C oracle` where /root/reference exists)')
    global _API
    try:
        api = pyref.hip_api(modules=True)
        import models
        import gangealing_amd.losses as L
        import gangealing_amd.stylegan2.networks as N
        import gangealing_amd.spatial_transformers.spatial_transformer as S
        assert models.Generator is N.Generator and models.get_stn is S.get_stn and models.gangealing_loss is L.gangealing_loss
        assert models.get_perceptual_loss is L.get_perceptual_loss
        assert models.DirectionInterpolator.__module__ == 'models.latent_learner'        # the reference's own
        assert os.path.samefile(os.path.dirname(os.path.dirname(models.__file__)), api.root)

        def run(api_):
            torch.manual_seed(0)
            gen = api_.Generator(64, 512, 8).to(cuda).eval().requires_grad_(False)
            kw = dict(flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=1)
            stn = api_.get_stn(['similarity', 'flow'], **kw).to(cuda)
            ema = api_.get_stn(['similarity', 'flow'], **kw).to(cuda)
            api_.accumulate(ema, stn, 0)
            ll = api_.DirectionInterpolator(None, 1, 5, gen.n_latent, 1).to(cuda)
            net = api_.LPIPS(net='vgg', lpips=False, pnet_rand=True, pretrained=False, verbose=False).to(cuda).eval()
            loss_fn = lambda x, y: net(x, y) / 18.0
            t_optim = torch.optim.Adam(stn.parameters(), lr=1e-4)
            ll_optim = torch.optim.Adam(ll.parameters(), lr=1e-3)
            losses = []
            for _ in range(3):
                ploss, delta = api_.gangealing_loss(gen, stn, ll, loss_fn, torch.nn.Sequential(), 0.5, 4, 512, False, cuda,
                                                    padding_mode='reflection')
                total = ploss + 1000.0 * api_.total_variation_loss(delta) + api_.flow_identity_loss(delta)
                stn.zero_grad()
                ll.zero_grad()
                total.backward()
                t_optim.step()
                ll_optim.step()
                api_.accumulate(ema, stn, 0.5 ** (32 / 10000))
                losses.append(float(total.detach()))
            return losses

        losses = run(api)
        assert all(np.isfinite(losses)), losses
        assert len(set(losses)) == 3                    # the parameters move
    finally:
        pyref.purge_models()                            # the other tests of this file bind the reference's own modules
        _API = None
