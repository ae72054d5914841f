"""GPU parity at BASELINE.json's own configurations, against vectors the REFERENCE produced on CPU
(oracle/make_golden_configs.py): Generator(256) and the similarity+flow STN at the benchmark shapes with batch 16
(the batch decides which tile variant of each convolution kernel is launched - 256-pixel patch tiles, the 128-wide
transposed tiles, the row-streaming weight gradient - so these compare exactly the kernels the benchmark runs with the
reference, not with each other), one full `gangealing_loss` step of C2 (batch 16, VGG loss form), the CelebA-HQ 512^2
flag set (C4) and the K=4 clustering objective with flips (C5), each in the exact-fp32 and the bf16x3 arithmetic.

The same driver (oracle/config_cases.run_config) that produced the fixtures from the reference's modules is called here
with gangealing_amd's modules.  Every comparison also records the error it measured; the session writes them to
gpurun_out/parity_report.json (committed per round as profiles/parity_rNN.json).

Tolerances.  north_star: fp32 activations within 1e-4.  Activations (images, flows, warped outputs) and every loss
term are asserted at 1e-4 (|err| <= 1e-4 * max(1, max|ref|); losses 1e-4 relative) in BOTH arithmetic modes.
Gradients are checked against a float64 evaluation of the reference, with the reference's own float32 error as the
yardstick and the relative L2 error as the metric (check_grads explains why not the largest entry).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, record_parity

pytestmark = pytest.mark.gpu

MODES = ['fp32', 'bf16x3', 'fp16x3']
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


GRAD_FACTOR = 4.0                        # HIP-path error allowed as a multiple of the reference's own fp32 error
# floors (relative L2 error of a gradient tensor, relative error of its norm).  They are set by "kink flips", not by
# rounding: a unit whose pre-activation lies within the forward rounding error of 0 takes the other ReLU / leaky-ReLU
# branch, and everything downstream of it inherits the change.  tests/test_gpu_lpips_masks.py demonstrates it on the
# VGG trunk: ONE flipped ReLU among 6.6 M units puts the input gradient 1.8e-3 (relative L2) from the reference, and
# with the reference's branch decisions pinned the same kernels reproduce it to 3.5e-6.  The reference's own float32
# run is subject to the same effect (its distance from float64 is 1e-3 .. 4e-3 on several parameters) and sometimes
# lucky (2e-6); the HIP path is not required to match that luck.  The kernels themselves are compared with float64
# convolutions at the same shapes, where nothing can flip, in tests/test_gpu_c2_layer_ops.py.
# Round 3: the library has no float atomics any more, so these errors are exactly reproducible run to run; the floors
# were re-measured (profiles/parity_r03.json: worst flow-stage / latent / generator gradient 9.7e-4 fp32, 3.3e-3
# bf16x3; perceptual-loss input gradient 1.8e-3 / 3.9e-3) and tightened from 5e-3 / 2e-2 to 3e-3 / 6e-3.
# fp16x3 (the benched arithmetic): its FORWARD is fp32-class, so the branch pattern - which is what these floors are about -
# is the exact-fp32 kernels'; it gets their floors (measured worst: flow stage 6.3e-4, latent 3.7e-4, generator 7e-4)
# Round 5: the library has been bitwise reproducible for two rounds, so the measured errors are the same on every box and
# run (profiles/parity_r05.json); the floors were re-measured over all 118 (fixture, stage, mode) records and lowered to
# <= 2x the worst: flow stage / generator fp32 9.2e-4, fp16x3 1.2e-3, bf16x3 3.1e-3; perceptual-loss input gradient
# 1.8e-3 (fp32: its one flipped ReLU) / 4.8e-6 / 3.95e-3; latent learner (own scale LL_FLOOR_SCALE) 2.8e-4 / 3.1e-4 / 7.7e-4
GRAD_FLOOR = {'fp32': (2.5e-3, 2.5e-3), 'bf16x3': (6e-3, 6e-3), 'fp16x3': (2e-3, 2e-3)}
LL_FLOOR_SCALE = 0.5
# the similarity stage additionally receives gradient through MipmapWarp's level selection, where a similarity warp
# makes the four neighbour distances EXACTLY tied in real arithmetic: arg-max (and with it the sub-gradient) is decided
# by last-ulp noise of the grid in every implementation, the reference's float32 and float64 runs included
# (measured worst case, c2_stn[1]: 7.9e-3 fp32 / 1.8e-2 bf16x3 against the reference's own 4.3e-3)
SIM_FACTOR = 8.0
# similarity-stage floors stay at round 3's 7.5e-3 (fp32, fp16x3) / 1.5e-2 (bf16x3), as multiples of GRAD_FLOOR (round 5
# measured worst outside c2_stn[1]: fp32 5.5e-3, fp16x3 7.9e-3 on cfg_c2t against the reference's own 1.1e-3, bf16x3 6.2e-3).
# These numbers bound DECISIONS, not arithmetic: tests/test_gpu_stn_decisions.py replays the reference's picks and gets
# 1.3e-5 / 2.5e-5; un-pinned, about half of the similarity warp's 16 384 arg-max picks are decided by the last ulp of the
# grid (exact ties in real arithmetic), so ANY change of rounding upstream re-draws them.  Measured in round 6
# (profiles/r06_h_similarity_stage_dice.txt, one box, this test under GG_DISABLE switches): folding ResBlock's 1 / sqrt(2)
# into its two branches - a few ulp on the trunk's activations - moves cfg_c2t from 5.5e-3 / 6.0e-3 / 5.9e-3 (fp32 / fp16x3
# / bf16x3) to 1.06e-2 / 3.1e-3 / 7.1e-3.  The fold is therefore NOT applied in the exact-product fp32 mode (it keeps the
# reference's operation order: networks.ResBlock.forward), and the split-precision modes' numbers stay inside these floors.
SIM_FLOOR_SCALE = {'fp32': 3.0, 'fp16x3': 3.75, 'bf16x3': 2.5}
# single entries, relative to the largest entry (round 5 measured worst: fp32 7.0e-3, fp16x3 9.0e-3, bf16x3 3.0e-2; one
# bound of 5e-2 for all modes before)
GRAD_MAX_ELEM = {'fp32': 1.5e-2, 'fp16x3': 2e-2, 'bf16x3': 5e-2}


def grad_errors(ours, ref32, ref64):
    """(relative L2 error of ours, of the reference's fp32 result, max-entry error of ours, of the reference) against
    the float64 gradient."""
    ref64 = np.asarray(ref64, dtype=np.float64)
    l2 = max(float(np.linalg.norm(ref64)), 1e-30)
    mx = max(float(np.abs(ref64).max()), 1e-30)
    d_o, d_r = np.asarray(ours, dtype=np.float64) - ref64, np.asarray(ref32, dtype=np.float64) - ref64
    return (float(np.linalg.norm(d_o)) / l2, float(np.linalg.norm(d_r)) / l2, float(np.abs(d_o).max()) / mx,
            float(np.abs(d_r).max()) / mx)


def check_grads(test, mode, grads, case, select=lambda name: True, factor=GRAD_FACTOR, floor_scale=1.0,
                sample_norms=False):
    """Gradient parity with a float64 evaluation of the REFERENCE as ground truth (fixture keys grad64_*).

    Metric: relative L2 error of each parameter's gradient (over the stored strided sample) and relative error of its
    norm, for the HIP path and for the reference's own float32 gradient.  Bound: GRAD_FACTOR x the reference's float32
    error, never tighter than GRAD_FLOOR.  The largest single-entry error is recorded too but only loosely bounded:
    the networks are piecewise linear (leaky ReLU / ReLU masks taken from the sign of an activation, max-pool and
    mip-level arg-max), so an activation within rounding distance of a kink can take the other branch in one
    implementation and moves a handful of gradient entries by O(|dy| |w|) - measured up to 2e-2 of the largest entry
    from a single flipped unit among 8.4 M, while the L2 error stays at 1e-5."""
    from oracle import config_cases as cc
    from conftest import PARITY
    norms, arrays = cc.pack_grads(grads)
    if sample_norms:          # the fixture's norms were taken over the stored strided samples (cfg_c5b16: halves averaged)
        norms = {k: float(np.linalg.norm(arrays['grad_' + k.replace('.', '_')].astype(np.float64))) for k in norms}
    meta = case['meta']
    names = [k for k in meta['grad_norms'] if select(k)]
    assert set(norms) >= set(names)
    if isinstance(floor_scale, dict):
        floor_scale = floor_scale[mode]
    n_floor, l2_floor = GRAD_FLOOR[mode][1] * floor_scale, GRAD_FLOOR[mode][0] * floor_scale
    rows, failures = [], []
    for name in names:
        key = name.replace('.', '_')
        l2_o, l2_r, mx_o, mx_r = grad_errors(arrays['grad_' + key], case['grad_' + key], case['grad64_' + key])
        n64 = max(meta['grad_norms64'][name], 1e-30)
        n_o = abs(norms[name] - n64) / n64
        n_r = abs(meta['grad_norms'][name] - n64) / n64
        rows.append((l2_o, l2_r, n_o, n_r, mx_o, mx_r, name))
        if l2_o > max(factor * l2_r, l2_floor) or n_o > max(factor * n_r, n_floor) or mx_o > GRAD_MAX_ELEM[mode]:
            failures.append((name, 'l2', l2_o, l2_r, 'norm', n_o, n_r, 'max', mx_o, mx_r))
    worst = sorted(rows, reverse=True)[:3]
    PARITY.setdefault(test, {}).setdefault(mode, {})['gradients_vs_reference_fp64'] = dict(
        params=len(rows),
        worst_rel_l2_err_ours=max(r[0] for r in rows), worst_rel_l2_err_reference_fp32=max(r[1] for r in rows),
        worst_norm_err_ours=max(r[2] for r in rows), worst_norm_err_reference_fp32=max(r[3] for r in rows),
        worst_max_entry_err_ours=max(r[4] for r in rows), worst_max_entry_err_reference_fp32=max(r[5] for r in rows),
        median_l2_ratio_ours_over_reference=float(np.median([r[0] / max(r[1], 1e-12) for r in rows])),
        worst_params=[dict(name=r[6], rel_l2_ours=r[0], rel_l2_reference_fp32=r[1]) for r in worst])
    assert not failures, failures[:4]


def check_one_grad(test, mode, tensor, ours, ref32, ref64):
    from conftest import PARITY
    l2_o, l2_r, mx_o, mx_r = grad_errors(ours, ref32, ref64)
    PARITY.setdefault(test, {}).setdefault(mode, {})[tensor + '_vs_reference_fp64'] = dict(
        rel_l2_err_ours=l2_o, rel_l2_err_reference_fp32=l2_r, max_entry_err_ours=mx_o, max_entry_err_reference_fp32=mx_r)
    assert l2_o <= max(GRAD_FACTOR * l2_r, GRAD_FLOOR[mode][0]) and mx_o <= GRAD_MAX_ELEM[mode], (l2_o, l2_r, mx_o, mx_r)


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
    # gradient of a 14-layer network w.r.t. its style input, against the reference evaluated in float64; the
    # reference's own float32 result sets the scale of what float32 arithmetic can deliver here
    check_one_grad('c2_generator', mode, 'gw', w.grad.cpu().numpy(), c['gw'], c['gw64'])


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
    x = cc.smooth_images(f'c2stn.x{ci}', n, 256, cuda)
    small = BilinearDownsample(2, 3).to(cuda)(x)
    out, flow = stn(small, return_flow=True, padding_mode=m['padding_mode'],
                    input_img_for_sampling=x if m['sample_from_full_res'] else None)
    test = f'c2_stn[{ci}]'
    # (image-like input, oracle/config_cases.smooth_images: on WHITE NOISE, where neighbouring pixels differ by ~0.7, a
    # flow error of 1e-6 - 1e-4 pixel - already moves the bilinear sample by 1e-4.)
    # Both arithmetic modes are held to the same 1e-4.  In bf16x3 the similarity trunk's FORWARD runs with three limbs
    # (conv_mfma.REGRESSION_PRECISION): with two, the four similarity parameters carried ~4e-5 relative error, which
    # moved the samples near the image corners by ~1e-3 pixel - 3.3e-4 on this textured input of amplitude 2
    # (round 2, and reproduced on CPU by scripts/study_bf16x3_stn.py, which also predicts 4.8e-5 for this form).
    check_batch(test, mode, out, c, 'out')
    check_batch(test, mode, flow, c, 'flow')
    gout = D(f'c2stn.g{ci}', tuple(out.shape), cuda)
    loss = (out * gout).mean() + 10.0 * total_variation_loss(flow) + flow_identity_loss(flow)
    err = record_parity(test, mode, 'loss', loss.detach().cpu().numpy(), c['loss'])
    assert err <= 1e-5 * max(1.0, abs(float(c['loss'])))
    params = list(stn.named_parameters())
    grads = torch.autograd.grad(loss, [p for _, p in params])
    # reported per stage: the similarity stage (stns.0) also receives gradient through MipmapWarp's level selection,
    # where a similarity warp makes the four neighbour distances exactly tied in real arithmetic
    named = {k: g for (k, _), g in zip(params, grads)}
    check_grads(test + '/flow-stage', mode, named, c, select=lambda k: k.startswith('stns.1.'))
    check_grads(test + '/similarity-stage', mode, named, c, select=lambda k: k.startswith('stns.0.'), factor=SIM_FACTOR,
                floor_scale=SIM_FLOOR_SCALE)


@pytest.mark.parametrize('name', ['c1', 'c2', 'c2t', 'c2r', 'c4', 'c5', 'c4b4', 'c5b4', 'c4b16', 'c5b8'])
def test_config_loss_step(name, mode, cuda):
    """One loss evaluation + backward of BASELINE config `name` (train.py:106-124) through the same driver that ran
    the reference.  c2t = c2 with textured generator images (per-pixel noise at full strength: neighbouring pixels of
    the warped 128^2 output differ by 0.07 on average at amplitude 2.5); c1 = configs[0], similarity-only STN (its
    "flow" is the (N, 2, 3) matrix and there are no flow regularisers).  Round 4: c2r = c2 with a wide dynamic range
    inside the generator (style layers drawn with scales 1e-5 .. 1e5: the operands of the modulated convolutions span
    1e-5 .. 3e5 from layer to layer - outside binary16's range in both directions; the block exponents of the fp16x3
    kernels have to hold 1e-4 there); c4 / c5 at the batch `bench.py --workload c4|c5` runs (4) and at the per-GPU
    batch of the reference's 8-GPU recipes (16; c5 at 8 - its float64 reference run at 16 exceeds the authoring container's memory): the batch decides which tile variant each convolution launches."""
    from oracle import config_cases as cc
    (c,) = load_golden(f'cfg_{name}')
    res = cc.run_config(our_api(), name, cuda)
    test = f'cfg_{name}'
    for key in ('unaligned', 'target', 'pred', 'stn_delta', 'delta_flow'):
        assert list(res[key].shape) == c['meta']['shapes'][key], (key, res[key].shape)
        check_batch(test, mode, res[key], c, key)
    has_flow = 'flow' in cc.CONFIGS[name]['transform']
    for key in ('ploss', 'total') + (('tv', 'identity') if has_flow else ()):
        err = record_parity(test, mode, key, res[key].cpu().numpy(), c[key])
        assert err <= 1e-4 * abs(float(c[key])) + 1e-9, (key, err, float(c[key]))       # every loss term to 1e-4 relative
    grads = res['grads']
    assert set(grads) == set(c['meta']['grad_norms'])
    if has_flow:
        check_grads(test + '/flow-stage', mode, grads, c, select=lambda k: k.startswith('stns.1.'))
        check_grads(test + '/similarity-stage', mode, grads, c, select=lambda k: k.startswith('stns.0.'),
                    factor=SIM_FACTOR, floor_scale=SIM_FLOOR_SCALE)
    else:       # a single similarity STN: its parameters carry no `stns.N.` prefix
        check_grads(test + '/similarity-stage', mode, grads, c, select=lambda k: k != 'll.coefficients',
                    factor=SIM_FACTOR, floor_scale=SIM_FLOOR_SCALE)
    check_grads(test + '/latent-learner', mode, grads, c, select=lambda k: k == 'll.coefficients', floor_scale=LL_FLOOR_SCALE)


def test_config_c2_one_limb_bf16_error_is_recorded(cuda):
    """BASELINE.json configs[1] names "bf16".  The package's plain-bf16 arithmetic (ONE bf16 limb per operand, fp32
    accumulate: `extras.bf16_eager` of the bench line) is not a parity mode - north_star's 1e-4 is met by fp16x3 - but its
    distance from the reference on the benchmark configuration is measured here and recorded (parity report ->
    profiles/parity_r06.json -> bench.py prints it as extras.bf16_eager.max_abs_err).  Bounds are sanity only: a limb of 8
    significand bits through 14 + 14 + 13 convolution layers leaves ~1e-2 of the activation scale."""
    from gangealing_amd.op import conv_mfma
    from oracle import config_cases as cc
    (c,) = load_golden('cfg_c2')
    old = conv_mfma.PRECISION
    conv_mfma.set_precision('bf16')
    try:
        res = cc.run_config(our_api(), 'c2', cuda)
    finally:
        conv_mfma.set_precision(old)
    worst = 0.0
    for key in ('unaligned', 'target', 'pred', 'stn_delta', 'delta_flow'):
        packed = cc.pack_batch(res[key], key)
        for part in ('first', 'sub'):
            k = f'{key}_{part}'
            err = record_parity('cfg_c2', 'bf16', k, packed[k], c[k])
            worst = max(worst, err / max(1.0, float(np.abs(c[k]).max())))
    for key in ('ploss', 'total', 'tv', 'identity'):
        record_parity('cfg_c2', 'bf16', key, res[key].cpu().numpy(), c[key])
    from conftest import PARITY
    PARITY['cfg_c2']['bf16']['worst_activation_err_over_scale'] = dict(value=worst)
    assert 1e-4 < worst < 0.25, worst          # far outside the parity bound, far inside "the same picture"
    assert abs(float(res['ploss']) - float(c['ploss'])) <= 0.2 * abs(float(c['ploss']))


def test_config_c5_batch16_from_reference_halves(mode, cuda):
    """C5 at the per-GPU batch `bench.py --workload c5 --batch 16` (extras.c5_batch16) and the reference's 8-GPU recipe
    run (scripts/training/lsun_cars.sh:4-7).  The reference cannot evaluate this batch on the authoring host (> 62 GB
    even in float32), so tests/golden/cfg_c5b16.npz holds its evaluation of the SAME sixteen samples in two halves
    (oracle/config_cases.py: CONFIGS['c5b16h0' / 'c5b16h1'], float32 and float64 each).  The step couples no samples -
    no batch statistics, every loss term a mean over samples - so the batch-16 run's activations are the halves' rows and
    its loss terms and gradients their averages; the HIP path runs the sixteen samples in ONE step (the tile variants a
    batch of 16 x 4 heads x 2 flips selects) and is held to that."""
    from oracle import config_cases as cc
    halves = load_golden('cfg_c5b16')
    assert len(halves) == 2
    cfg = cc.CONFIGS['c5b16']
    res = cc.run_config(our_api(), 'c5b16', cuda)
    test = 'cfg_c5b16'
    for k, c in enumerate(halves):
        part = tuple(c['meta']['cfg']['part'])
        for key in ('unaligned', 'target', 'pred', 'stn_delta', 'delta_flow'):
            rows = cc.sample_rows(res[key], cfg, part)
            assert list(rows.shape) == c['meta']['shapes'][key], (key, rows.shape)
            check_batch(f'{test}/half{k}', mode, rows, c, key)
    for key in ('ploss', 'total', 'tv', 'identity'):
        ref = 0.5 * (float(halves[0][key]) + float(halves[1][key]))
        err = record_parity(test, mode, key, res[key].cpu().numpy(), np.float64(ref))
        assert err <= 1e-4 * abs(ref) + 1e-9, (key, err, ref)
    # gradients: the average of the halves' (strided samples are taken at the same positions; norms over the samples)
    both = {}
    for key in halves[0]:
        if key.startswith('grad_') or key.startswith('grad64_'):
            both[key] = 0.5 * (halves[0][key].astype(np.float64) + halves[1][key].astype(np.float64))
    names = list(halves[0]['meta']['grad_norms'])
    both['meta'] = dict(grad_norms={n: float(np.linalg.norm(both['grad_' + n.replace('.', '_')])) for n in names},
                        grad_norms64={n: float(np.linalg.norm(both['grad64_' + n.replace('.', '_')])) for n in names})
    grads = res['grads']
    assert set(grads) == set(names)
    check_grads(test + '/flow-stage', mode, grads, both, select=lambda k: k.startswith('stns.1.'), sample_norms=True)
    check_grads(test + '/similarity-stage', mode, grads, both, select=lambda k: k.startswith('stns.0.'),
                factor=SIM_FACTOR, floor_scale=SIM_FLOOR_SCALE, sample_norms=True)
    check_grads(test + '/latent-learner', mode, grads, both, select=lambda k: k == 'll.coefficients', sample_norms=True,
                floor_scale=LL_FLOOR_SCALE)


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
    check_one_grad(test, mode, 'gin0', in0.grad.cpu().numpy(), case['gin0'], case['gin0_64'])
