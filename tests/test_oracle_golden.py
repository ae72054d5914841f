"""CPU suite: the numpy oracle (oracle/np_ops.py) against the golden vectors that
oracle/make_golden.py produced from the reference's own CPU bodies."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import np_ops as O


def close(a, b, atol, rtol=1e-5):
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


@pytest.mark.parametrize('case', load_golden('upfirdn2d'), ids=lambda c: c['meta']['tag'])
def test_upfirdn2d(case):
    m = case['meta']
    up, down, pad = (m['up'],) * 2, (m['down'],) * 2, tuple(m['pad'])
    out = O.upfirdn2d(case['x'], case['k'], up, down, pad)
    assert out.shape == case['out'].shape
    close(out, case['out'], 2e-6)
    gx = O.upfirdn2d_backward(case['g'], case['k'], up, down, pad, case['x'].shape)
    assert gx.shape == case['x'].shape
    close(gx, case['gx'], 2e-6)


@pytest.mark.parametrize('case', load_golden('fused_act'))
def test_fused_act(case):
    out = O.fused_leaky_relu(case['x'], case['b'])
    close(out, case['out'], 1e-6)
    gx, gb = O.fused_leaky_relu_backward(case['g'], out)
    # sign reference is the OUTPUT (fused_act.py:60): identical to autograd except where x+b == 0
    nz = (case['x'] + case['b'].reshape([1, -1] + [1] * (case['x'].ndim - 2))) != 0
    close(gx[nz], case['gx'][nz], 1e-6)
    if nz.all():
        close(gb, case['gb'], 1e-5)


@pytest.mark.parametrize('case', load_golden('grid_sample'), ids=lambda c: c['meta']['padding_mode'] + '-' + c['meta']['grid'])
def test_grid_sample(case):
    out = O.grid_sample(case['x'], case['grid'], case['meta']['padding_mode'])
    close(out, case['out'], 2e-6)


@pytest.mark.parametrize('case', load_golden('mipmap_warp') + load_golden('mipmap_warp_deep'),
                         ids=lambda c: f"{c['x'].shape[-1]}-{c['meta']['padding_mode']}-{c['meta']['grid']}")
def test_mipmap_warp(case):
    m = case['meta']
    out, aux = O.mipmap_warp(case['x'], case['grid'], m['max_num_levels'], 0.0, m['padding_mode'], return_aux=True)
    close(out, case['out'], 5e-6)
    close(aux['levels'] / (m['max_num_levels'] - 1.0), case['levels_map'], 1e-6)
    # the integer floor/ceil levels from the exponent trick == floor/ceil of the reference's levels
    ref_levels = case['levels_map'] * np.float32(m['max_num_levels'] - 1.0)
    _, dmax = O.mip_levels(case['grid'], case['x'].shape[2], case['x'].shape[3], m['max_num_levels'])
    fl, ce = O.mip_level_ints(dmax, m['max_num_levels'])
    safe = np.abs(ref_levels - np.round(ref_levels)) > 1e-5        # reference value not within rounding of an integer
    assert (fl[safe] == np.floor(ref_levels[safe])).all()
    assert (ce[safe] == np.ceil(ref_levels[safe])).all()
    assert (fl == aux['level_floor']).all() and (ce == aux['level_ceil']).all()


def test_similarity_head():
    (c,) = load_golden('similarity_head')
    m = O.make_affine_matrix(c['params'])
    close(m, c['matrix'], 1e-6)
    comp = O.compose_affine(c['base'], m)
    close(comp, c['composed'], 1e-6)
    grid = O.affine_grid(comp, 16, 16)
    close(grid, c['grid'], 2e-6)


@pytest.mark.parametrize('case', load_golden('flow_head'))
def test_flow_head(case):
    close(O.identity_flow(case['identity'].shape[1]), case['identity'], 1e-7)
    flow, delta = O.flow_compose(case['low'], case['mask'], case['base'], 8)
    close(delta, case['delta'], 2e-6)
    close(flow, case['flow'], 2e-6)
    res = O.interpolate_bilinear(np.ascontiguousarray(case['flow'].transpose(0, 3, 1, 2)), 2.0).transpose(0, 2, 3, 1)
    close(res, case['resized2x'], 2e-6)


@pytest.mark.parametrize('case', load_golden('bilinear_downsample'))
def test_bilinear_downsample(case):
    close(O.bilinear_downsample(case['x'], case['meta']['stride']), case['out'], 2e-6)


@pytest.mark.parametrize('case', load_golden('flow_losses'))
def test_flow_losses(case):
    close(O.total_variation_loss(case['delta']), case['tv'], 1e-6)
    close(O.flow_identity_loss(case['delta']), case['identity'], 1e-6)


@pytest.mark.parametrize('case', load_golden('modulated_conv'))
def test_modulated_conv(case):
    m = case['meta']
    scale = 1.0 / np.sqrt(case['mod_weight'].shape[1])
    style = case['w'] @ (case['mod_weight'] * np.float32(scale)).T + case['mod_bias']
    close(style, case['style'], 1e-5)
    blur = (np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0 * 4).astype(np.float32)
    out = O.modulated_conv2d(case['x'], case['weight'], case['style'], m['demodulate'], m['upsample'], blur, (1, 1))
    close(out, case['out'], 2e-5, 1e-4)


@pytest.mark.parametrize('case', load_golden('splat2d'))
def test_splat2d_restatement_matches_reference_kernel(case):
    """oracle/np_ops.splat2d (restated from splat_gpu_impl.cu:53-95, splat_gpu.c:14-41) against outputs of the
    reference's own kernel run on an MI355X (oracle/make_golden_splat.py)."""
    assert case['meta']['source'] == 'reference-kernel'
    out = O.splat2d(case['input'], case['coords'], case['values'], case['sigma'], case['meta']['soft_normalize'])
    close(out, case['out'], 2e-6, 1e-5)          # the kernel sums with float atomics in arbitrary order
    assert np.isfinite(out).all()


# ------------------------------------------------------------------ model-level oracle (torch_ref) vs reference

def _det_sd(module, rules=()):
    from oracle.det_weights import det_state_dict
    return det_state_dict(module, [tuple(r) for r in rules])


def test_torch_ref_generator_matches_reference():
    import torch
    from oracle import torch_ref as R
    from gangealing_amd.stylegan2 import Generator
    (c,) = load_golden('generator16')
    g = Generator(16, 512, 8)
    sd = _det_sd(g)
    noise = [torch.from_numpy(c[f'noise{i}']) for i in range(c['meta']['num_layers'])]
    with torch.no_grad():
        img, latent = R.generator(sd, torch.from_numpy(c['z']), 16, noise)
    close(latent[:, 0].numpy(), c['w'], 1e-5)
    close(img.numpy(), c['img'], 2e-4, 1e-4)
    w = torch.from_numpy(c['w']).requires_grad_(True)
    img2 = R.generator_synthesis(sd, w.unsqueeze(1).repeat(1, g.n_latent, 1), 16, noise)
    img2.backward(torch.from_numpy(c['gimg']))
    close(w.grad.numpy(), c['gw'], 2e-3, 1e-3)


@pytest.mark.parametrize('case', load_golden('stn'), ids=lambda c: '+'.join(c['meta']['transforms']))
def test_torch_ref_stn_matches_reference(case):
    import torch
    from oracle import torch_ref as R
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    m = case['meta']
    stn = get_stn(m['transforms'], flow_size=m['flow_size'], supersize=m['supersize'], channel_multiplier=0.5, num_heads=1)
    sd = {k: v.clone().requires_grad_(True) for k, v in _det_sd(stn, m['scale_rules']).items()}
    x = torch.from_numpy(case['x'])
    src = x if m['supersize'] > m['flow_size'] else None
    out, fm = R.composed_stn(sd, x, m['flow_size'], m['supersize'], m['padding_mode'], tuple(m['transforms']), source=src)
    close(out.detach().numpy(), case['out'], 2e-4, 1e-4)
    close(fm.detach().numpy(), case['flow_or_matrix'], 2e-4, 1e-4)
    loss = (out ** 2).mean()
    if 'flow' in m['transforms']:
        loss = loss + 10.0 * R.total_variation_loss(fm) + fm.pow(2).mean()
    close(loss.detach().numpy(), case['loss'], 1e-5, 1e-4)
    loss.backward()
    for name, ref_norm in m['grad_norms'].items():
        if ref_norm is None:
            continue
        got = float(sd[name].grad.double().norm())
        assert abs(got - ref_norm) <= 2e-3 * max(ref_norm, 1e-6) + 1e-7, (name, got, ref_norm)


def test_torch_ref_train_step_matches_reference():
    import torch
    from oracle import torch_ref as R
    from oracle.det_weights import det_array
    from gangealing_amd.stylegan2 import Generator
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    (c,) = load_golden('train_step')
    m = c['meta']
    g_sd = _det_sd(Generator(m['gen_size'], 512, 8))
    stn = get_stn(['similarity', 'flow'], flow_size=m['flow_size'], supersize=m['gen_size'], channel_multiplier=0.5, num_heads=1)
    stn_sd = {k: v.clone().requires_grad_(True) for k, v in _det_sd(stn, m['scale_rules']).items()}
    T = lambda name, shape, s=1.0: torch.from_numpy(det_array(name, shape, s))
    ll_sd = dict(directions=T('ll.directions', (m['ndirs'], 512)), lat_mean=T('ll.lat_mean', (1, 512)),
                 coefficients=T('ll.coefficients', (1, m['ndirs']), 0.3).requires_grad_(True))
    nl = (int(np.log2(m['gen_size'])) - 2) * 2 + 1
    res = lambda i: 2 ** ((i + 5) // 2)
    n1 = [T(f'ts.n1.{i}', (2, 1, res(i), res(i))) for i in range(nl)]
    n2 = [T(f'ts.n2.{i}', (2, 1, res(i), res(i))) for i in range(nl)]
    total, parts = R.train_loss(g_sd, stn_sd, ll_sd, torch.from_numpy(c['z']), m['gen_size'], m['flow_size'], m['psi'],
                                m['inject'], m['padding_mode'], ('similarity', 'flow'), R.mse_loss_fn,
                                m['tv_weight'], m['flow_identity_weight'], n1, n2)
    close(parts['unaligned'].numpy(), c['unaligned'], 5e-4, 1e-4)
    close(parts['pred'].detach().numpy(), c['pred'], 5e-4, 1e-4)
    close(total.detach().numpy(), c['total'], 1e-4, 1e-4)
    total.backward()
    close(ll_sd['coefficients'].grad.numpy(), c['g_coefficients'], 1e-4, 2e-3)
    worst = 0.0
    for name, ref_norm in m['grad_norms'].items():
        if name == 'll.coefficients':
            continue
        got = float(stn_sd[name].grad.double().norm())
        worst = max(worst, abs(got - ref_norm) / max(ref_norm, 1e-9))
    assert worst < 5e-3, worst


@pytest.mark.parametrize('case', load_golden('cluster_classifier'), ids=lambda c: f"heads{c['meta']['num_heads']}")
def test_torch_ref_cluster_classifier_matches_reference(case):
    """ResnetClassifier: logits, cross-entropy gradients and the reverse top-k accuracy of the
    oracle restatement against the reference module's own outputs."""
    import torch
    from oracle import torch_ref as R
    from gangealing_amd.cluster_classifier import ResnetClassifier
    m = case['meta']
    net = ResnetClassifier(m['size'], channel_multiplier=m['channel_multiplier'], num_heads=m['num_heads'],
                           supersize=m['supersize'])
    sd = {k: v.clone().requires_grad_(True) for k, v in _det_sd(net, m['scale_rules']).items()}
    logits = R.cluster_classifier(sd, torch.from_numpy(case['x']), m['size'])
    close(logits.detach().numpy(), case['logits'], 2e-4, 1e-4)
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(case['labels']))
    close(loss.detach().numpy(), case['loss'], 1e-5, 1e-4)
    loss.backward()
    close(sd['to_logits.weight'].grad.numpy(), case['grad_to_logits_weight'], 1e-5, 2e-3)
    for name, ref_norm in m['grad_norms'].items():
        got = float(sd[name].grad.double().norm())
        assert abs(got - ref_norm) <= 2e-3 * max(ref_norm, 1e-6) + 1e-7, (name, got, ref_norm)
    scores = torch.from_numpy(case['scores'])
    assert float(R.reverse_topk_accuracy(logits, scores)) == float(case['acc1'])
    assert float(R.reverse_topk_accuracy(logits, scores, k=2)) == float(case['acc2'])


def test_max_pool_restatement_matches_aten():
    """np_ops.max_pool2x2 (+ backward) against F.max_pool2d on the CPU - the operator torchvision's VGG16 runs in the
    reference's perceptual loss - incl. exact ties inside windows, NaNs and odd sizes."""
    import torch
    import torch.nn.functional as F
    from oracle import np_ops
    rs = np.random.RandomState(3)
    for shape in [(2, 3, 8, 8), (1, 2, 7, 9), (2, 1, 2, 2)]:
        x = (rs.randn(*shape) * 2).round().astype(np.float32) / 2
        x[0, 0, 0, 1] = np.nan
        xt = torch.from_numpy(x).requires_grad_(True)
        ref = F.max_pool2d(xt, 2, 2)
        g = rs.randn(*ref.shape).astype(np.float32)
        ref.backward(torch.from_numpy(g))
        out, code = np_ops.max_pool2x2(x, return_code=True)
        np.testing.assert_array_equal(np.nan_to_num(out, nan=123.0), np.nan_to_num(ref.detach().numpy(), nan=123.0))
        np.testing.assert_array_equal(np_ops.max_pool2x2_backward(g, code, shape[2:]), xt.grad.numpy())
