"""Decision replay for the WHOLE spatial transformer (round 6) - what test_gpu_act_masks.py left un-pinned.

tests/test_gpu_configs.py bounds the similarity stage's gradients at 7.5e-3 (relative L2 against the reference's float64
evaluation), five times looser than the flow stage, with the argument that MipmapWarp's mip-level arg-max (antialiased_
sampling.py:62-97: torch.max over the distances to the four neighbours, the level's sub-gradient flows to the winner) has
exactly tied candidates under a similarity warp, so the winner is last-ulp noise in every implementation.  This file
replaces the argument by a replay.  tests/golden/stn_decisions.npz (oracle/make_golden_configs.py stn_decisions) holds,
for the STN run of act_masks (similarity + flow at 64^2, batch 4), the reference's own
  * arg-max index of both anti-aliased warps,
  * output sign of the similarity stage's final EqualLinear(activation='fused_lrelu') (an (N, C) function call),
  * output signs of the RAFT head's two plain ReLUs (warping_heads.py:160-169),
and act_masks.npz its leaky-ReLU decisions and float32 / float64 gradients.  With ALL of them pinned the HIP path must
reproduce the reference's float32 gradient of EVERY parameter - similarity stage included - to rounding accuracy; the
number of decisions that differ un-pinned is counted.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, PARITY

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['fp32', 'fp16x3'])
def mode(request):
    from gangealing_amd.op import conv_mfma
    old = conv_mfma.PRECISION
    conv_mfma.set_precision(request.param)
    yield request.param
    conv_mfma.set_precision(old)
    conv_mfma.ACT_OBSERVER = None


def bits(arr, shape, device):
    return torch.from_numpy(np.unpackbits(arr)[:int(np.prod(shape))].reshape(shape).astype(bool)).to(device)


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


class Pinner:
    """conv_mfma.ACT_OBSERVER for 4-D (conv trunk) AND 2-D (final EqualLinear) leaky-ReLU outputs, in call order each."""

    def __init__(self, signs4, signs2, pin):
        self.signs = {4: list(signs4), 2: list(signs2)}
        self.k = {4: 0, 2: 0}
        self.pin, self.flips, self.units = pin, {4: 0, 2: 0}, 0

    def __call__(self, site, y):
        d = y.dim()
        ref = self.signs[d][self.k[d]]
        assert tuple(ref.shape) == tuple(y.shape), (site, tuple(ref.shape), tuple(y.shape))
        self.k[d] += 1
        data = y.data
        ours = data > 0
        self.units += ref.numel()
        self.flips[d] += int((ours != ref).sum())
        if self.pin:
            data[ref & ~ours] = 1e-30
            data[~ref & ours] = -1e-30


def test_every_stn_gradient_is_the_references_once_all_decisions_are_pinned(mode, cuda):
    from gangealing_amd.op import conv_mfma
    from gangealing_amd.spatial_transformers import antialiased_sampling as aa
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    from gangealing_amd.losses import total_variation_loss, flow_identity_loss
    from oracle import config_cases as cc
    from test_gpu_configs import load_det, D
    from test_gpu_act_masks import unpack_signs
    case = load_golden('act_masks')[1]
    (dec,) = load_golden('stn_decisions')
    m, dm = case['meta'], dec['meta']
    n = m['batch']
    signs4 = unpack_signs(case, cuda)
    signs2 = [bits(dec[f'head_sign{k}'], shp, cuda) for k, shp in enumerate(dm['head_shapes'])]
    relus = [bits(dec[f'relu{k}'], shp, cuda) for k, shp in enumerate(dm['relu_shapes'])]
    level_args = [torch.from_numpy(dec[f'level_arg{k}'].astype(np.int8)).to(cuda) for k in range(2)]
    names = list(m['grad_norms'])
    out = {}
    for pin in (False, True):
        stn = load_det(get_stn(['similarity', 'flow'], flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=1),
                       cc.STN_RULES).to(cuda)
        x = cc.smooth_images('actmask.stn.x', n, 64, cuda)
        pinner = Pinner(signs4, signs2, pin)
        relu_state = dict(k=0, flips=0)

        def relu_hook(mod, inp, o):
            ref = relus[relu_state['k']]
            relu_state['k'] += 1
            ours = o > 0
            relu_state['flips'] += int((ours != ref).sum())
            if pin:                                   # ATen's threshold_backward reads the OUTPUT
                o = o.clone()
                o.data[ref & ~ours] = 1e-30
                o.data[~ref & ours] = 0.0
                return o
        hooks = [mod.register_forward_hook(relu_hook) for mod in stn.modules() if isinstance(mod, torch.nn.ReLU)]
        # our own arg-max decisions, for the count: evaluated on the grids the two warps are called with
        seen_grids = []
        real_apply = aa._MipmapWarpFn.apply

        def spy(inputs, grid, *a):
            seen_grids.append((grid.detach(), inputs.shape[-2], inputs.shape[-1]))
            return real_apply(inputs, grid, *a)
        aa._MipmapWarpFn.apply = staticmethod(spy)
        conv_mfma.ACT_OBSERVER = pinner
        aa.LEVEL_ARG_PINS = [t for t in level_args] if pin else None
        try:
            warped, flow = stn(x, return_flow=True, padding_mode=m['padding_mode'])
        finally:
            conv_mfma.ACT_OBSERVER = None
            aa.LEVEL_ARG_PINS = None
            aa._MipmapWarpFn.apply = real_apply
            for h in hooks:
                h.remove()
        assert pinner.k[4] == len(signs4) and pinner.k[2] == len(signs2) and relu_state['k'] == len(relus)
        assert len(seen_grids) == 2
        arg_flips = []
        for (grid, h, w), ref_arg in zip(seen_grids, level_args):
            ours = aa.warp_level_arg(grid, h, w)
            arg_flips.append(int((ours != ref_arg.to(torch.int32)).sum()))
        loss = (warped * D('actmask.stn.g', tuple(warped.shape), cuda)).mean() + 10.0 * total_variation_loss(flow) + \
            flow_identity_loss(flow)
        assert abs(float(loss.detach()) - float(case['loss'])) <= 1e-5 * max(1.0, abs(float(case['loss'])))
        params = list(stn.named_parameters())
        grads = torch.autograd.grad(loss, [p for _, p in params])
        _, arrays = cc.pack_grads({k: g_ for (k, _), g_ in zip(params, grads)})
        per_stage = {}
        for stage in ('stns.0.', 'stns.1.'):
            errs = [(rel_l2(arrays['grad_' + k.replace('.', '_')], case['grad_' + k.replace('.', '_')]), k)
                    for k in names if k.startswith(stage)]
            per_stage[stage] = max(errs)
        out['pinned' if pin else 'free'] = dict(
            lrelu_flips=pinner.flips[4], head_flips=pinner.flips[2], relu_flips=relu_state['flips'],
            level_arg_flips=arg_flips, level_arg_points=[int(t.numel()) for t in level_args],
            similarity_stage_worst_rel_l2_vs_reference_fp32=per_stage['stns.0.'][0],
            similarity_stage_worst_param=per_stage['stns.0.'][1],
            flow_stage_worst_rel_l2_vs_reference_fp32=per_stage['stns.1.'][0],
            flow_stage_worst_param=per_stage['stns.1.'][1])
    out['reference_fp32_vs_fp64_similarity_stage_worst'] = max(
        rel_l2(case['grad_' + k.replace('.', '_')], case['grad64_' + k.replace('.', '_')])
        for k in names if k.startswith('stns.0.'))
    PARITY.setdefault('stn_decisions', {})[mode] = out
    # with every decision the reference took, the reference's gradient - of the similarity stage too
    bound = 1e-4
    assert out['pinned']['similarity_stage_worst_rel_l2_vs_reference_fp32'] <= bound, out
    assert out['pinned']['flow_stage_worst_rel_l2_vs_reference_fp32'] <= bound, out
    # and the un-pinned distance of the similarity stage is explained by decisions, not arithmetic
    assert out['pinned']['similarity_stage_worst_rel_l2_vs_reference_fp32'] <= \
        out['free']['similarity_stage_worst_rel_l2_vs_reference_fp32'] * 1.001 + 1e-7, out
