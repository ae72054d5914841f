"""Decision replay for the generator's and the STN's leaky ReLUs (round 5; the counterpart of test_gpu_lpips_masks.py).

The gradient tolerances of tests/test_gpu_configs.py rest on one claim: the networks are piecewise linear, an activation
whose pre-activation lies within forward rounding of zero takes the other branch in another implementation, and everything
downstream inherits that - so gradients sit ~1e-3 (not ~1e-6) from the reference's although the backward arithmetic is
exact to rounding.  tests/golden/act_masks.npz holds the REFERENCE's own branch decisions - one bit per FusedLeakyReLU
unit, in call order - for a Generator(64) run and a similarity + flow STN run, with the reference's float32 and float64
gradients of the same runs (oracle/make_golden_configs.py act_masks).  Here the HIP path

  1. counts how many of its own decisions differ from the reference's, and
  2. re-runs with the reference's decisions PINNED (conv_mfma.ACT_OBSERVER hands every activation output to the test
     right after it is produced; a flipped unit is moved to the reference's side of the kink by a 1e-30-sized edit of the
     tensor the backward reads; no sign plane is kept meanwhile) and must then reproduce the reference's float32 gradient
     to rounding accuracy - far below the un-pinned distance wherever a consequential flip exists.

Not pinned: the (N, C) activation of the similarity head's EqualLinear (a function call, not a module, in the reference),
the ReLUs of the RAFT flow head, and MipmapWarp's mip-level arg-max (exactly tied candidates under a similarity warp) -
the similarity stage's parameters, which receive gradient through that arg-max, are therefore only reported."""
import numpy as np
import pytest
import torch

from conftest import load_golden, PARITY

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['fp32', 'fp16x3', 'bf16x3'])
def mode(request):
    from gangealing_amd.op import conv_mfma
    old = conv_mfma.PRECISION
    conv_mfma.set_precision(request.param)
    yield request.param
    conv_mfma.set_precision(old)
    conv_mfma.ACT_OBSERVER = None


def unpack_signs(case, device):
    out = []
    for k, shp in enumerate(case['meta']['shapes']):
        bits = np.unpackbits(case[f'sign{k:02d}'])[:int(np.prod(shp))].reshape(shp).astype(bool)
        out.append(torch.from_numpy(bits).to(device))
    return out


class Pinner:
    """conv_mfma.ACT_OBSERVER: compares (and, when `pin`, aligns) the sign of every 4-D activation output with the
    reference's decision for the same layer, in call order."""

    def __init__(self, signs, pin):
        self.signs, self.pin = signs, pin
        self.k = self.flips = self.units = 0
        self.sites = []

    def __call__(self, site, y):
        if y.dim() != 4:                 # the similarity head's EqualLinear activation: not recorded by the fixture
            return
        assert self.k < len(self.signs), f'more activation layers than the reference recorded ({site})'
        ref = self.signs[self.k]
        assert tuple(ref.shape) == tuple(y.shape), (self.k, site, tuple(ref.shape), tuple(y.shape))
        self.k += 1
        self.sites.append(site)
        data = y.data
        ours = data > 0
        self.units += ref.numel()
        self.flips += int((ours != ref).sum())
        if self.pin:
            data[ref & ~ours] = 1e-30            # the backward tests `> 0` on this tensor
            data[~ref & ours] = -1e-30


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def test_generator_gradient_is_exact_once_lrelu_decisions_are_pinned(mode, cuda):
    from gangealing_amd.op import conv_mfma
    from gangealing_amd.stylegan2 import Generator
    from test_gpu_configs import load_det, D
    case = load_golden('act_masks')[0]
    m = case['meta']
    n = m['batch']
    g = load_det(Generator(64, 512, 8)).to(cuda).eval().requires_grad_(False)
    noise = [D(f'actmask.gen.noise{i}', (n, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)), cuda) for i in range(g.num_layers)]
    signs = unpack_signs(case, cuda)
    out = {}
    for pin in (False, True):
        pinner = Pinner(signs, pin)
        conv_mfma.ACT_OBSERVER = pinner
        w = torch.from_numpy(case['w']).to(cuda).requires_grad_(True)
        img, _ = g([w.unsqueeze(1).repeat(1, g.n_latent, 1)], input_is_latent=True, noise=noise)
        conv_mfma.ACT_OBSERVER = None
        assert pinner.k == len(signs), (pinner.k, len(signs), pinner.sites)
        err = float((img.detach().cpu() - torch.from_numpy(case['img'])).abs().max())
        assert err <= 1e-4 * max(1.0, float(np.abs(case['img']).max())), err       # pinning does not move the forward
        (gw,) = torch.autograd.grad(img, w, D('actmask.gen.gimg', tuple(img.shape), cuda))
        out['pinned' if pin else 'free'] = dict(rel_l2_vs_reference_fp32=rel_l2(gw.cpu().numpy(), case['gw']),
                        rel_l2_vs_reference_fp64=rel_l2(gw.cpu().numpy(), case['gw64']),
                        flips=pinner.flips, units=pinner.units)
    out['reference_fp32_vs_fp64'] = rel_l2(case['gw'], case['gw64'])
    PARITY.setdefault('act_masks[generator]', {})[mode] = out
    # only a handful of the 11 M units sit close enough to a kink to flip
    assert out['free']['flips'] <= (64 if mode != 'bf16x3' else 512), out
    # with the reference's decisions, the reference's gradient: rounding of nine modulated layers (the data gradients of
    # fp16x3 run on binary16 limbs, of bf16x3 on bf16 limbs)
    bound = {'fp32': 2e-5, 'fp16x3': 2e-5, 'bf16x3': 2e-4}[mode]
    assert out['pinned']['rel_l2_vs_reference_fp32'] <= bound, out
    assert out['pinned']['rel_l2_vs_reference_fp32'] <= out['free']['rel_l2_vs_reference_fp32'] * 1.001 + 1e-7, out


def test_stn_gradients_are_exact_once_lrelu_decisions_are_pinned(mode, cuda):
    from gangealing_amd.op import conv_mfma
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    from gangealing_amd.losses import total_variation_loss, flow_identity_loss
    from oracle import config_cases as cc
    from test_gpu_configs import load_det, D
    case = load_golden('act_masks')[1]
    m = case['meta']
    n = m['batch']
    signs = unpack_signs(case, cuda)
    names = list(m['grad_norms'])
    out = {}
    for pin in (False, True):
        stn = load_det(get_stn(['similarity', 'flow'], flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=1),
                       cc.STN_RULES).to(cuda)
        x = cc.smooth_images('actmask.stn.x', n, 64, cuda)
        pinner = Pinner(signs, pin)
        conv_mfma.ACT_OBSERVER = pinner
        warped, flow = stn(x, return_flow=True, padding_mode=m['padding_mode'])
        conv_mfma.ACT_OBSERVER = None
        assert pinner.k == len(signs), (pinner.k, len(signs), pinner.sites)
        loss = (warped * D('actmask.stn.g', tuple(warped.shape), cuda)).mean() + 10.0 * total_variation_loss(flow) + \
            flow_identity_loss(flow)
        assert abs(float(loss) - float(case['loss'])) <= 1e-5 * max(1.0, abs(float(case['loss'])))
        params = list(stn.named_parameters())
        grads = torch.autograd.grad(loss, [p for _, p in params])
        _, arrays = cc.pack_grads({k: g_ for (k, _), g_ in zip(params, grads)})
        per_stage = {}
        for stage in ('stns.0.', 'stns.1.'):
            errs = [(rel_l2(arrays['grad_' + k.replace('.', '_')], case['grad_' + k.replace('.', '_')]), k)
                    for k in names if k.startswith(stage)]
            per_stage[stage] = max(errs)
        out['pinned' if pin else 'free'] = dict(flips=pinner.flips, units=pinner.units,
                        flow_stage_worst_rel_l2_vs_reference_fp32=per_stage['stns.1.'][0],
                        flow_stage_worst_param=per_stage['stns.1.'][1],
                        similarity_stage_worst_rel_l2_vs_reference_fp32=per_stage['stns.0.'][0],
                        similarity_stage_worst_param=per_stage['stns.0.'][1])
    ref = max(rel_l2(case['grad_' + k.replace('.', '_')], case['grad64_' + k.replace('.', '_')])
              for k in names if k.startswith('stns.1.'))
    out['reference_fp32_vs_fp64_flow_stage_worst'] = ref
    PARITY.setdefault('act_masks[stn]', {})[mode] = out
    assert out['free']['flips'] <= (256 if mode != 'bf16x3' else 2048), out
    # flow stage: every leaky-ReLU decision of its trunk is pinned; what is left un-pinned on its gradient path are the two
    # plain ReLUs of the RAFT head (warping_heads.py:130-135) and the similarity stage's output it consumes
    bound = {'fp32': 1e-4, 'fp16x3': 1e-4, 'bf16x3': 1e-3}[mode]
    assert out['pinned']['flow_stage_worst_rel_l2_vs_reference_fp32'] <= bound, out
