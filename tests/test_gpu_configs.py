"""GPU parity at BASELINE.json's own configurations, against vectors the REFERENCE produced on CPU
(oracle/make_golden_configs.py): Generator(256) and the similarity+flow STN at the benchmark shapes with batch 16
(the batch decides which tile variant of each convolution kernel is launched - 256-pixel patch tiles, the 128-wide
transposed tiles, the row-streaming weight gradient - so these compare exactly the kernels the benchmark runs with the
reference, not with each other), one full `gangealing_loss` step of C2 (batch 16, VGG loss form), the CelebA-HQ 512^2
flag set (C4) and the K=4 clustering objective with flips (C5), each in the exact-fp32 and the bf16x3 arithmetic.

The same driver (oracle/config_cases.run_config) that produced the fixtures from the reference's modules is called here
with gangealing_amd's modules.  Every comparison also records the error it measured; the session writes them to
gpurun_out/parity_report.json (committed per round as profiles/parity_rNN.json).

Tolerances.  north_star: fp32 activations within 1e-4.  Activations (images, flows, warped outputs) are asserted at
|err| <= 1e-4 * max(1, max|ref|) in BOTH arithmetic modes.  Gradients are compared relative to the largest entry of
each tensor; the bounds are stated next to each assert and the measured values are in the report.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, record_parity

pytestmark = pytest.mark.gpu

MODES = ['fp32', 'bf16x3']
ACT_TOL = 1e-4


@pytest.fixture(params=MODES)
def mode(request):
    from gangealing_amd.op import conv_mfma
    old = conv_mfma.PRECISION
    conv_mfma.set_precision(request.param)
    yield request.param
    conv_mfma.set_precision(old)


def our_api():
    from oracle import config_cases as cc
    from gangealing_amd.stylegan2 import Generator
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    from gangealing_amd.spatial_transformers.antialiased_sampling import BilinearDownsample
    from gangealing_amd.latent_learner import DirectionInterpolator
    from gangealing_amd.losses import (LPIPS, gangealing_loss, gangealing_cluster_loss, total_variation_loss,
                                       flow_identity_loss)
    return cc.api_namespace(Generator=Generator, get_stn=get_stn, BilinearDownsample=BilinearDownsample,
                            DirectionInterpolator=DirectionInterpolator, LPIPS=LPIPS, gangealing_loss=gangealing_loss,
                            gangealing_cluster_loss=gangealing_cluster_loss, total_variation_loss=total_variation_loss,
                            flow_identity_loss=flow_identity_loss)


def check_batch(test, mode, got, case, prefix, tol=ACT_TOL):
    """Compare the stored slices of a (N, ...) tensor: first sample in full, strided subsample of all, per-sample sums."""
    from oracle import config_cases as cc
    packed = cc.pack_batch(got, prefix)
    worst = 0.0
    for part in ('first', 'sub'):
        key = f'{prefix}_{part}'
        err = record_parity(test, mode, key, packed[key], case[key])
        scale = max(1.0, float(np.abs(case[key]).max()))
        worst = max(worst, err / scale)
        assert err <= tol * scale, (key, err, scale)
    # per-sample sums in float64: an error anywhere in a sample moves them (bound: tol * sum |x|)
    s_err = np.abs(packed[f'{prefix}_sum'] - case[f'{prefix}_sum'])
    bound = tol * np.maximum(case[f'{prefix}_abssum'], 1.0)
    record_parity(test, mode, f'{prefix}_sum', packed[f'{prefix}_sum'], case[f'{prefix}_sum'])
    assert (s_err <= bound).all(), (prefix, s_err.max(), bound.min())
    return worst


def check_grads(test, mode, grads, case, norm_tol, elem_tol, skip=lambda name: False):
    """Per-parameter gradient norms (relative) and the stored strided samples (relative to the tensor's largest
    entry).  -> (worst norm error, worst element error) over the compared parameters."""
    from oracle import config_cases as cc
    norms, arrays = cc.pack_grads(grads)
    ref_norms = case['meta']['grad_norms']
    assert set(norms) == set(ref_norms), set(norms) ^ set(ref_norms)
    worst_n = worst_e = 0.0
    per_param = {}
    for name, ref_norm in ref_norms.items():
        key = 'grad_' + name.replace('.', '_')
        n_err = abs(norms[name] - ref_norm) / max(ref_norm, 1e-12)
        scale = float(np.abs(case[key]).max())
        e_err = float(np.abs(arrays[key] - case[key]).max()) / max(scale, 1e-20)
        per_param[name] = (n_err, e_err)
        if skip(name):
            continue
        worst_n, worst_e = max(worst_n, n_err), max(worst_e, e_err)
    from conftest import PARITY
    PARITY.setdefault(test, {}).setdefault(mode, {})['gradients'] = dict(
        worst_norm_rel_err=worst_n, worst_elem_err_rel_to_max=worst_e,
        worst_params=sorted(((max(v), k) for k, v in per_param.items()), reverse=True)[:4])
    assert worst_n <= norm_tol, ('grad norm', worst_n, sorted(((v[0], k) for k, v in per_param.items()), reverse=True)[:3])
    assert worst_e <= elem_tol, ('grad element', worst_e, sorted(((v[1], k) for k, v in per_param.items()), reverse=True)[:3])
    return worst_n, worst_e


def load_det(module, rules=()):
    from oracle.det_weights import det_state_dict
    torch.nn.Module.load_state_dict(module, det_state_dict(module, [tuple(r) for r in rules]), strict=False)
    return module


def D(name, shape, device, scale=1.0):
    from oracle.det_weights import det_array
    return torch.from_numpy(det_array(name, shape, scale)).to(device)


def test_c2_generator_batch16(mode, cuda):
    """Generator(256) at batch 16: image, and the gradient w.r.t. w through all 14 style inputs."""
    from gangealing_amd.stylegan2 import Generator
    (c,) = load_golden('c2_generator')
    n = c['meta']['batch']
    g = load_det(Generator(256, 512, 8)).to(cuda).eval().requires_grad_(False)
    noise = [D(f'c2gen.noise{i}', (n, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)), cuda) for i in range(g.num_layers)]
    with torch.no_grad():
        img, latent = g([torch.from_numpy(c['z']).to(cuda)], return_latents=True, noise=noise)
    err = record_parity('c2_generator', mode, 'w', latent[:, 0].cpu().numpy(), c['w'])
    assert err <= 1e-5
    check_batch('c2_generator', mode, img, c, 'img')
    w = torch.from_numpy(c['w']).to(cuda).requires_grad_(True)
    img2, _ = g([w.unsqueeze(1).repeat(1, g.n_latent, 1)], input_is_latent=True, noise=noise)
    check_batch('c2_generator', mode, img2, c, 'img_from_w')
    img2.backward(D('c2gen.gimg', tuple(img2.shape), cuda))
    err = record_parity('c2_generator', mode, 'gw', w.grad.cpu().numpy(), c['gw'])
    # gradient of a 14-layer network w.r.t. its style input: compared relative to the largest entry
    assert err <= (2e-3 if mode == 'fp32' else 1e-2) * float(np.abs(c['gw']).max()), err


@pytest.mark.parametrize('ci', [0, 1], ids=['resized-reflection', 'fullres-border'])
def test_c2_stn_batch16(ci, mode, cuda):
    """similarity+flow STN, regression at 128^2 from a 256^2 image, batch 16: warped output, flow, all gradients."""
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    from gangealing_amd.spatial_transformers.antialiased_sampling import BilinearDownsample
    from gangealing_amd.losses import total_variation_loss, flow_identity_loss
    from oracle import config_cases as cc
    c = load_golden('c2_stn')[ci]
    m = c['meta']
    n = m['batch']
    stn = get_stn(['similarity', 'flow'], flow_size=128, supersize=256, channel_multiplier=0.5, num_heads=1)
    stn = load_det(stn, cc.STN_RULES).to(cuda)
    x = D(f'c2stn.x{ci}', (n, 3, 256, 256), cuda, 0.5)
    small = BilinearDownsample(2, 3).to(cuda)(x)
    out, flow = stn(small, return_flow=True, padding_mode=m['padding_mode'],
                    input_img_for_sampling=x if m['sample_from_full_res'] else None)
    test = f'c2_stn[{ci}]'
    check_batch(test, mode, out, c, 'out')
    check_batch(test, mode, flow, c, 'flow')
    gout = D(f'c2stn.g{ci}', tuple(out.shape), cuda)
    loss = (out * gout).mean() + 10.0 * total_variation_loss(flow) + flow_identity_loss(flow)
    err = record_parity(test, mode, 'loss', loss.detach().cpu().numpy(), c['loss'])
    assert err <= 1e-5 * max(1.0, abs(float(c['loss'])))
    params = list(stn.named_parameters())
    grads = torch.autograd.grad(loss, [p for _, p in params])
    # the similarity stage (stns.0) receives part of its gradient through MipmapWarp's level selection, where a
    # similarity warp makes the four neighbour distances exactly tied in real arithmetic and arg-max is decided by
    # last-ulp noise of the grid: its gradients are recorded and bounded separately
    flow_stage = lambda name: name.startswith('stns.1.')
    check_grads(test + '/flow-stage', mode, {k: g for (k, _), g in zip(params, grads) if flow_stage(k)},
                dict(c, meta=dict(m, grad_norms={k: v for k, v in m['grad_norms'].items() if flow_stage(k)})),
                norm_tol=2e-3 if mode == 'fp32' else 1e-2, elem_tol=5e-3 if mode == 'fp32' else 2e-2)
    check_grads(test + '/similarity-stage', mode, {k: g for (k, _), g in zip(params, grads) if not flow_stage(k)},
                dict(c, meta=dict(m, grad_norms={k: v for k, v in m['grad_norms'].items() if not flow_stage(k)})),
                norm_tol=3e-2, elem_tol=1e-1)


@pytest.mark.parametrize('name', ['c2', 'c4', 'c5'])
def test_config_loss_step(name, mode, cuda):
    """One loss evaluation + backward of BASELINE config `name` (train.py:106-124) through the same driver that ran
    the reference."""
    from oracle import config_cases as cc
    (c,) = load_golden(f'cfg_{name}')
    res = cc.run_config(our_api(), name, cuda)
    test = f'cfg_{name}'
    for key in ('unaligned', 'target', 'pred', 'stn_delta', 'delta_flow'):
        assert list(res[key].shape) == c['meta']['shapes'][key], (key, res[key].shape)
        check_batch(test, mode, res[key], c, key)
    for key in ('ploss', 'tv', 'identity', 'total'):
        err = record_parity(test, mode, key, res[key].cpu().numpy(), c[key])
        assert err <= 1e-4 * max(1.0, abs(float(c[key]))) if key != 'total' else True, (key, err)
    rel_total = abs(float(res['total']) - float(c['total'])) / abs(float(c['total']))
    assert rel_total <= 1e-4, rel_total
    grads = res['grads']
    flow_stage = lambda k: k.startswith('stns.1.') or k == 'll.coefficients'
    norms = c['meta']['grad_norms']
    check_grads(test + '/flow-stage+ll', mode, {k: g for k, g in grads.items() if flow_stage(k)},
                dict(c, meta=dict(c['meta'], grad_norms={k: v for k, v in norms.items() if flow_stage(k)})),
                norm_tol=2e-3 if mode == 'fp32' else 1e-2, elem_tol=5e-3 if mode == 'fp32' else 2e-2)
    check_grads(test + '/similarity-stage', mode, {k: g for k, g in grads.items() if not flow_stage(k)},
                dict(c, meta=dict(c['meta'], grad_norms={k: v for k, v in norms.items() if not flow_stage(k)})),
                norm_tol=3e-2, elem_tol=1e-1)


@pytest.mark.parametrize('case', load_golden('lpips'), ids=lambda c: 'lin' if c['meta']['lpips'] else 'baseline')
def test_lpips_golden(case, mode, cuda):
    """The perceptual loss in both forms against the reference LPIPS class run on a VGG16 with the same weights."""
    from oracle import config_cases as cc
    from gangealing_amd.losses import LPIPS
    lp = case['meta']['lpips']
    net = LPIPS(net='vgg', lpips=lp, pnet_rand=True, pretrained=False)
    torch.nn.Module.load_state_dict(net, cc.det_lpips_state_dict(net), strict=False)
    net = net.to(cuda).eval()
    in0 = torch.from_numpy(case['in0']).to(cuda).requires_grad_(True)
    in1 = torch.from_numpy(case['in1']).to(cuda)
    val, per_layer = net(in0, in1, retPerLayer=True)
    test = f"lpips[{'lin' if lp else 'baseline'}]"
    err = record_parity(test, mode, 'val', val.detach().cpu().numpy(), case['val'])
    assert err <= 1e-4 * max(1.0, float(np.abs(case['val']).max()))
    got_layers = torch.cat([p.reshape(3, 1) for p in per_layer], 1).detach().cpu().numpy()
    err = record_parity(test, mode, 'per_layer', got_layers, case['per_layer'])
    assert err <= 1e-4 * max(1.0, float(np.abs(case['per_layer']).max()))
    val.backward(torch.from_numpy(case['g']).to(cuda))
    err = record_parity(test, mode, 'gin0', in0.grad.cpu().numpy(), case['gin0'])
    assert err <= (2e-3 if mode == 'fp32' else 1e-2) * float(np.abs(case['gin0']).max())
