"""GPU tests of the split-precision convolution path (bf16 matrix pipe, 2 or 3 bf16 limbs per fp32
operand, fp32 accumulation): against the exact-fp32 MFMA kernel and against the reference goldens."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture
def precision():
    from gangealing_amd.op import conv_mfma
    old = conv_mfma.PRECISION
    yield conv_mfma.set_precision
    conv_mfma.set_precision(old)


CASES = [
    # n, cin, cout, h, k, stride, pad, mode, scaled
    (2, 64, 128, 33, 3, 1, 1, 0, True),
    (3, 128, 64, 16, 3, 1, 1, 0, False),
    (2, 64, 96, 17, 3, 2, 0, 0, False),       # strided correlation
    (2, 96, 64, 9, 3, 2, 0, 1, True),         # transposed stride 2 (parity classes)
    (1, 512, 512, 4, 3, 1, 1, 0, True),       # split-K + atomics
    (2, 64, 130, 20, 1, 1, 0, 0, False),      # 1x1, ragged cout
    (2, 128, 64, 15, 1, 2, 0, 1, False),      # 1x1 transposed: odd positions stay zero
    # all-parity-classes transposed kernel (power-of-two input width >= 16)
    (2, 64, 130, (16, 16), 3, 2, 0, 1, True),     # 16-wide tiles, ragged cout
    (2, 64, 128, (21, 32), 3, 2, 0, 1, True),     # 32-wide tiles, ragged tile rows
    (1, 96, 64, (33, 64), 3, 2, 1, 1, False),     # pad 1 (data gradient of a stride-2 conv) + output_padding
    (4, 32, 128, (128, 128), 3, 2, 0, 1, True),   # enough tiles for the 128-q (8-wave) variant
    (1, 256, 256, (16, 16), 3, 2, 0, 1, True),    # split-K + atomics
    (2, 64, 96, (4, 4), 3, 2, 0, 1, True),        # 4-wide image: one tile covers the whole q-grid
    (3, 64, 64, (7, 8), 3, 2, 1, 1, False),       # 8-wide, pad 1
    # 3x3 / stride 2 / pad 0 patch-reuse tile (conv_s2_patch.hip: output width a multiple of 32, height of 4)
    (2, 64, 96, (65, 65), 3, 2, 0, 0, True),      # 32^2 outputs, ragged cout, style + demodulation scales
    (3, 96, 160, (17, 129), 3, 2, 0, 0, False),   # 8 x 64 outputs, 6 chunks of 16 channels, two cout tiles (one ragged)
    (1, 512, 256, (9, 65), 3, 2, 0, 0, True),     # a single pixel tile: split-K over the channel chunks
    (2, 64, 64, (66, 130), 3, 2, 0, 0, False),    # even input sizes (the last input row / column is never read)
    (5, 32, 128, (33, 193), 3, 2, 0, 0, True),    # 16 x 96 outputs: three tile columns, two chunks
]


@pytest.mark.parametrize('mode_name,tol', [('bf16x3', 3e-5), ('bf16x6', 1e-5), ('fp16x3', 1e-5)])
@pytest.mark.parametrize('spec', CASES, ids=lambda s: 'x'.join(map(str, s)))
def test_split_conv_matches_fp32_kernel(spec, mode_name, tol, cuda, precision):
    from gangealing_amd.op import conv_mfma as cm
    n, cin, cout, h, k, stride, pad, mode, scaled = spec
    g = torch.Generator(device='cpu').manual_seed(1234)
    hh, ww = h if isinstance(h, tuple) else (h, h + 2)
    x = torch.randn(n, cin, hh, ww, generator=g).to(cuda)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(cuda)
    s_in = (torch.rand(n, cin, generator=g) + 0.5).to(cuda) if scaled else None
    s_out = (torch.rand(n, cout, generator=g) + 0.5).to(cuda) if scaled else None
    bias = torch.randn(cout, generator=g).to(cuda)
    pw = cm.PackedWeight(w, 1, cout, cin, k, 0, 0, 0.7)
    assert pw.split_ok()
    out_hw = (2 * hh, 2 * ww) if (mode == 1 and pad == 1) else None       # output_padding = 1
    precision('fp32')
    ref = cm.conv_forward(x, pw, n, 1, cin, cout, k, stride, pad, mode, in_scale=s_in, out_scale=s_out, bias=bias,
                          out_hw=out_hw)
    precision(mode_name)
    out = cm.conv_forward(x, pw, n, 1, cin, cout, k, stride, pad, mode, in_scale=s_in, out_scale=s_out, bias=bias,
                          out_hw=out_hw)
    assert out.shape == ref.shape
    err = float((out - ref).abs().max() / ref.abs().max())
    assert err < tol, err


def test_ineligible_layers_stay_on_fp32_kernel(cuda, precision):
    from gangealing_amd.op import conv_mfma as cm
    assert not cm.PackedWeight(torch.zeros(64, 3, 3, 3, device=cuda), 1, 64, 3, 3, 0, 0).split_ok()     # 3-channel stem
    assert not cm.PackedWeight(torch.zeros(3, 128, 1, 1, device=cuda), 1, 3, 128, 1, 0, 0).split_ok()   # ToRGB
    x = torch.randn(2, 3, 8, 8, device=cuda)
    w = torch.randn(64, 3, 3, 3, device=cuda)
    precision('fp32')
    a = cm.conv2d(x, w, padding=1)
    precision('bf16x3')
    b = cm.conv2d(x, w, padding=1)
    assert torch.equal(a, b)


@pytest.mark.parametrize('mode_name', ['bf16x3', 'bf16x6'])
def test_generator_and_train_step_golden_under_split_precision(mode_name, cuda, precision):
    """Forward parity with the REFERENCE (1e-4 class) also holds on the split-precision path."""
    import test_gpu_models as M
    precision(mode_name)
    # activations (the 1e-4 gate) use the same tolerances as the fp32 kernel; the GRADIENT checks are
    # cancellation-heavy reductions, so the 2-limb mode (2^-16 per product) gets 10x looser bounds there
    loose = mode_name == 'bf16x3'
    M.test_generator_golden(cuda, grad_tol=(2e-2, 1e-2) if loose else (2e-3, 1e-3))
    M.test_train_step_golden(cuda, grad_rel=5e-2 if loose else 5e-3)


def test_autograd_through_split_conv(cuda, precision):
    import torch.nn.functional as F
    from gangealing_amd.op import conv2d_gradfix
    precision('bf16x3')
    x = torch.randn(2, 64, 12, 12, device=cuda, requires_grad=True)
    w = (torch.randn(96, 64, 3, 3, device=cuda) / 24).requires_grad_(True)
    out = conv2d_gradfix.conv2d(x, w, padding=1)
    g = torch.randn_like(out)
    out.backward(g)
    xc, wc = x.detach().cpu().requires_grad_(True), w.detach().cpu().requires_grad_(True)
    F.conv2d(xc, wc, padding=1).backward(g.cpu())
    np.testing.assert_allclose(x.grad.cpu().numpy(), xc.grad.numpy(), atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(w.grad.cpu().numpy(), wc.grad.numpy(), atol=5e-4, rtol=5e-4)   # wgrad stays fp32 MFMA


@pytest.mark.parametrize('mode_name', ['fp32', 'bf16x3', 'bf16x6'])
@pytest.mark.parametrize('shape', [(2, 64, 96, 16, 0), (1, 512, 512, 4, 0), (4, 32, 128, 64, 0), (16, 64, 128, 64, 0),
                                   (2, 64, 96, 16, 1), (3, 32, 64, 40, 1), (2, 32, 64, 32, 1), (1, 32, 64, 128, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_styled_conv_fused_activation_matches_unfused(shape, mode_name, cuda, precision):
    """One-kernel StyledConv tail: the conv epilogue (or, on up-sampling layers, the blur kernel) carries noise +
    bias + leaky ReLU, and the backward applies the leaky-ReLU mask while staging the adjoint blur's input."""
    from gangealing_amd.op.fused_act import noise_bias_leaky_relu
    from gangealing_amd.stylegan2.networks import StyledConv
    n, cin, cout, res, up = shape
    precision(mode_name)
    torch.manual_seed(7)
    layer = StyledConv(cin, cout, 3, 32, upsample=bool(up)).to(cuda).requires_grad_(False)
    layer.noise.weight.fill_(0.37)
    layer.activate.bias.copy_(torch.randn(cout, device=cuda) * 0.5)
    x = torch.randn(n, cin, res, res, device=cuda, requires_grad=True)
    style = torch.randn(n, 32, device=cuda)
    noise = torch.randn(n, 1, res * (2 if up else 1), res * (2 if up else 1), device=cuda)
    assert layer.conv.can_fuse_act(x, style, layer.noise.weight, layer.activate.bias)
    fused = layer(x, style, noise=noise)
    g = torch.randn_like(fused)
    (dx_fused,) = torch.autograd.grad(fused, x, g)
    pre = layer.conv(x, style)
    ref = noise_bias_leaky_relu(pre, noise, layer.noise.weight, layer.activate.bias, layer.activate.negative_slope,
                                layer.activate.scale)
    (dx_ref,) = torch.autograd.grad(ref, x, g)
    scale = float(ref.detach().abs().max())
    # same kernels, same accumulation order: only the epilogue's fused multiply-add differs
    assert float((fused - ref).detach().abs().max()) <= 2e-6 * scale
    # The two backward passes take their leaky-ReLU masks from their OWN forward outputs, which differ by ~1e-7: an
    # activation that close to zero takes the other branch in one of them (measured: one such unit among 4.2 M in the
    # 256^2 case moves a 3x3 neighbourhood of dx by 5e-3 of the largest entry).  Without a flip the gradients agree to
    # rounding; with flips only the mean-square error is bounded.
    flips = int(((fused > 0) != (ref > 0)).sum())
    err = (dx_fused - dx_ref).abs()
    if flips == 0:
        assert float(err.max()) <= 2e-6 * float(dx_ref.abs().max())
    else:
        assert flips <= 4, flips
        assert float(err.norm() / dx_ref.norm()) <= 2e-4 * flips ** 0.5, (flips, float(err.max()))
        assert float((err > 2e-6 * float(dx_ref.abs().max())).float().mean()) <= 1e-3 * flips
    # a style that needs a gradient (learned W+ slots) must keep the unfused path
    assert not layer.conv.can_fuse_act(x, style.clone().requires_grad_(True), layer.noise.weight, layer.activate.bias)


@pytest.mark.parametrize('mode_name,tol', [('bf16x3', 3e-5), ('bf16x6', 1e-5)])
@pytest.mark.parametrize('spec', [(2, 64, 64, 32, 32, 1), (1, 48, 72, 40, 64, 1), (3, 160, 136, 9, 32, 1),
                                  (2, 64, 192, 16, 96, 2), (16, 64, 64, 128, 128, 1),
                                  (3, 96, 160, 16, 16, 1), (2, 64, 64, 6, 16, 1), (16, 512, 512, 16, 16, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_row_streaming_wgrad_matches_fp32_kernel(spec, mode_name, tol, cuda, precision):
    """3x3 / stride 1 / pad 1 weight gradient (conv3x3_wgrad_rows_kernel: rolling x window, alignbit-shifted taps)
    against the exact-fp32 MFMA kernel; width % 32 == 0 selects it, other shapes keep the generic kernel."""
    from gangealing_amd.op import conv_mfma as cm
    n, cin, cout, h, w, groups = spec
    g = torch.Generator(device='cpu').manual_seed(99)
    x = torch.randn(n, cin * groups, h, w, generator=g).to(cuda)
    dy = torch.randn(n, cout * groups, h, w, generator=g).to(cuda)
    precision('fp32')
    ref = cm.conv_wgrad(x, dy, n, groups, cin, cout, 3, 1, 1, 0.5)
    precision(mode_name)
    out = cm.conv_wgrad(x, dy, n, groups, cin, cout, 3, 1, 1, 0.5)
    assert out.shape == ref.shape == (groups * cout, cin, 3, 3)
    err = float((out - ref).abs().max() / ref.abs().max())
    assert err < tol, err
    # accumulate-into form: adds to what is there
    slot = torch.ones_like(ref)
    cm.conv_wgrad(x, dy, n, groups, cin, cout, 3, 1, 1, 0.5, into=slot)
    assert float((slot - 1 - ref).abs().max() / ref.abs().max()) < tol


@pytest.mark.parametrize('mode_name', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('shape', [(2, 64, 64, 32, 32), (1, 128, 96, 16, 16), (3, 32, 64, 8, 12), (1, 512, 512, 4, 4),
                                   (2, 96, 160, 24, 64), (4, 64, 64, 128, 128)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_conv3x3_bias_act_matches_unfused(shape, mode_name, cuda, precision):
    """conv + bias + leaky ReLU in the convolution's epilogue (STN ConvLayer / VGG conv+ReLU) == conv, then
    fused_leaky_relu; forward, data gradient, weight gradient and bias gradient."""
    from gangealing_amd.op import conv_mfma as cm
    from gangealing_amd.op.fused_act import fused_leaky_relu
    n, cin, cout, h, w = shape
    precision(mode_name)
    g = torch.Generator(device='cpu').manual_seed(11)
    x = torch.randn(n, cin, h, w, generator=g).to(cuda).requires_grad_(True)
    wt = torch.randn(cout, cin, 3, 3, generator=g).to(cuda).requires_grad_(True)
    b = torch.randn(cout, generator=g).to(cuda).requires_grad_(True)
    scale = (cin * 9) ** -0.5
    for alpha, gain, opt_in in ((0.2, 2 ** 0.5, ()), (0.0, 1.0, ()), (0.2, 2 ** 0.5, ('mask_wgrad',))):
        cm.ENABLED = frozenset(opt_in)           # opt-in: activation backward inside both gradient kernels
        fused = cm.conv3x3_bias_act(x, wt, b, alpha, gain, weight_scale=scale)
        go = torch.randn_like(fused)
        gf = torch.autograd.grad(fused, (x, wt, b), go)
        ref = fused_leaky_relu(cm.conv2d(x, wt, None, 1, 1, weight_scale=scale), b, alpha, gain)
        gr = torch.autograd.grad(ref, (x, wt, b), go)
        assert float((fused - ref).detach().abs().max()) <= 2e-6 * float(ref.detach().abs().max())
        cm.ENABLED = frozenset()
        for a_, r_ in zip(gf, gr):
            assert float((a_ - r_).abs().max()) <= 1e-5 * float(r_.abs().max()) + 1e-7


def test_masked_dgrad_serves_patch_shapes_only(cuda, precision):
    """gg_conv3x3_masked_dgrad_f32 fuses the leaky-ReLU backward into the data gradient where the patch-reuse kernel
    runs, and reports "not served" (no launch) elsewhere; the result equals lrelu-backward followed by the conv."""
    from gangealing_amd.op import conv_mfma as cm
    from gangealing_amd import _lib
    precision('bf16x3')
    g = torch.Generator(device='cpu').manual_seed(21)
    w = (torch.randn(64, 96, 3, 3, generator=g) / 30).to(cuda)            # layer: 96 -> 64 channels
    pw = cm.PackedWeight(w, 1, 96, 64, 3, 1, 1, 1.0)                       # data-gradient pack (reduce over 64)
    for res, served in ((32, True), (4, False)):
        dy = torch.randn(2, 64, res, res, generator=g).to(cuda)
        y = torch.randn(2, 64, res, res, generator=g).to(cuda)
        dx = cm.masked_dgrad(dy, y, 0.2, 2 ** 0.5, pw, 2, 64, 96, res, res)
        assert (dx is not None) == served
        if served:
            gm = torch.empty_like(dy)
            _lib.call('gg_fused_lrelu_bwd_f32', gm, None, dy, y, 0.2, 2 ** 0.5, 2, 64, res * res)
            ref = cm.conv_forward(gm, pw, 2, 1, 64, 96, 3, 1, 1, 0)
            assert float((dx - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    precision('fp32')
    assert cm.masked_dgrad(dy, y, 0.2, 1.0, pw, 2, 64, 96, 4, 4) is None


@pytest.mark.parametrize('mode_name', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('spec', [(16, 3, 64, 128, 128), (2, 3, 70, 8, 8), (3, 1, 16, 16, 32), (2, 4, 130, 24, 8)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_pointwise_small_cin_wgrad(spec, mode_name, cuda, precision):
    """1x1 weight gradient of the few-input-channel stems (streaming reduction kernel) against float64 einsum."""
    from gangealing_amd.op import conv_mfma as cm
    n, cin, cout, h, w = spec
    precision(mode_name)
    g = torch.Generator(device='cpu').manual_seed(17)
    x = torch.randn(n, cin, h, w, generator=g)
    dy = torch.randn(n, cout, h, w, generator=g)
    ref = 0.25 * torch.einsum('nohw,nihw->oi', dy.double(), x.double())
    out = cm.conv_wgrad(x.to(cuda), dy.to(cuda), n, 1, cin, cout, 1, 1, 0, 0.25)
    assert out.shape == (cout, cin, 1, 1)
    np.testing.assert_allclose(out.cpu().numpy().reshape(cout, cin), ref.numpy(), rtol=2e-5, atol=2e-5 * float(ref.abs().max()))
    slot = torch.full((cout, cin, 1, 1), 2.0, device=cuda)
    cm.conv_wgrad(x.to(cuda), dy.to(cuda), n, 1, cin, cout, 1, 1, 0, 0.25, into=slot)
    np.testing.assert_allclose(slot.cpu().numpy().reshape(cout, cin) - 2.0, ref.numpy(), rtol=2e-5,
                               atol=2e-5 * float(ref.abs().max()))


RANGE_CASES = [
    # n, cin, cout, (h, w), k, stride, pad, mode, in_scale - one per kernel family that stages binary16 limbs
    (4, 128, 128, (32, 64), 3, 1, 1, 0, True),     # conv3x3_patch_kernel, 256-pixel (8-wave, pipelined) tile
    (2, 96, 64, (16, 16), 3, 1, 1, 0, False),      # conv3x3_patch_kernel, 128-pixel tile, 64-channel outputs
    (2, 64, 96, (17, 19), 3, 2, 0, 0, False),      # conv_split_kernel: strided correlation
    (2, 64, 130, (20, 22), 1, 1, 0, 0, True),      # conv_split_kernel: 1x1
    (4, 64, 128, (128, 128), 3, 2, 0, 1, True),    # convT3x3s2_patch_kernel, 128-q tile
    (2, 96, 128, (21, 32), 3, 2, 0, 1, False),     # convT3x3s2_patch_kernel, 64-q tile
    (1, 512, 512, (4, 4), 3, 1, 1, 0, True),       # split-K: every split carries its own exponent
    (2, 64, 96, (65, 67), 3, 2, 0, 0, True),       # conv3x3s2_patch_kernel (conv_s2_patch.hip): 32 x 32 outputs
]


def _range_case(spec, cuda, seed=99):
    from gangealing_amd.op import conv_mfma as cm
    n, cin, cout, (hh, ww), k, stride, pad, mode, scaled = spec
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(n, cin, hh, ww, generator=g).to(cuda)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(cuda)
    s_in = (torch.rand(n, cin, generator=g) + 0.5).to(cuda) if scaled else None
    pw = cm.PackedWeight(w, 1, cout, cin, k, 0, 0, 1.0)
    run = lambda xx, grad=False: cm.conv_forward(xx, pw, n, 1, cin, cout, k, stride, pad, mode, in_scale=s_in, grad=grad)
    return x, run


@pytest.mark.parametrize('scale', [1e-30, 1e-12, 1e-6, 1.0, 3e5, 1e7, 1e30], ids=lambda v: f'{v:g}')
@pytest.mark.parametrize('spec', RANGE_CASES, ids=lambda s: 'x'.join(map(str, s)))
def test_fp16_block_exponent_any_magnitude(spec, scale, cuda, precision):
    """Round 4: binary16 limbs carry a per-tile block exponent taken from the staged data (csrc/conv_mfma.hip,
    BlockExp), so the fp16x3 kernels hold fp32-class accuracy (<= 1e-5 of the output's largest entry; measured 2e-6)
    for operands of ANY magnitude - 1e-30 .. 1e30 here, forward and data-gradient launches - where round 3 lost limb 0
    to subnormals below 6e-5, saturated above 65504 and returned inf beyond 1.3e5.  Reference: the exact-product fp32
    MFMA kernel on the same scaled input (scaling by the test's decimal factors is not exact, so it is NOT ref * s)."""
    x, run = _range_case(spec, cuda)
    xs = x * scale
    precision('fp32')
    ref = run(xs)
    assert bool(torch.isfinite(ref).all()) and float(ref.abs().max()) > 0
    from gangealing_amd.op import conv_mfma as cm
    precision('fp16x3')
    for grad in (False, True):
        out = run(xs, grad)
        assert bool(torch.isfinite(out).all())
        err = float((out - ref).abs().max() / ref.abs().max())
        # gradient launches of shapes served by the generic re-gathering kernel keep bf16 limbs (conv_mfma.limb_code):
        # the two-bf16-limb bound, at any magnitude as well
        n, cin, cout, (hh, ww), k, stride, pad, mode, scaled = spec
        bf16 = grad and cm.limb_code(True, cm._generic_shape(k, stride, pad, mode, ww, hh)) == 2
        assert err <= (3e-5 if bf16 else 1e-5), (grad, err)


@pytest.mark.parametrize('spec', RANGE_CASES[:6] + RANGE_CASES[7:], ids=lambda s: 'x'.join(map(str, s)))
def test_fp16_block_exponent_grows_inside_a_tile(spec, cuda, precision):
    """Chunks of 32 input channels (and image regions) of very different magnitude inside ONE launch: the tile's exponent
    has to grow while it accumulates (accumulators rescaled by an exact power of two) or stay put when a later chunk is
    small.  Both channel orders, plus one image quadrant 1e6 times larger than the rest (tiles with different
    exponents side by side).  Error bound relative to the largest output entry, as for any dot product."""
    x, run = _range_case(spec, cuda, seed=5)
    cin = x.shape[1]
    steps = [1e-7, 1.0, 3e4, 1e-3, 7e5, 1e-12]
    ramp = torch.tensor([steps[(c // 32) % len(steps)] for c in range(cin)], device=x.device).view(1, cin, 1, 1)
    quadrant = torch.ones_like(x[:1, :1])
    quadrant[..., : x.shape[-2] // 2, : x.shape[-1] // 2] = 1e6
    for name, xs in (('ascending', x * ramp), ('descending', x * ramp.flip(1)), ('quadrant', x * quadrant),
                     ('zero-chunks', x * (ramp >= 1.0))):
        precision('fp32')
        ref = run(xs)
        precision('fp16x3')
        out = run(xs)
        assert bool(torch.isfinite(out).all()), name
        err = float((out - ref).abs().max() / ref.abs().max())
        assert err <= 1e-5, (name, err)
    # The exponent belongs to a TILE: where the quiet region's tiles do not touch the loud quadrant (the two cases whose
    # images span many tiles) it is judged on its own scale.  Inside one tile accuracy is relative to the tile's largest
    # operand (2^-31 of it per element) - an image that fits a single tile, like the 16^2 cases here, cannot resolve a
    # 1e6 : 1 range between neighbouring regions to 1e-5 of the quiet one, and fp32-class parity (1e-4 of the tensor's
    # scale) does not ask for it.
    sl = (slice(None), slice(None), slice(out.shape[-2] * 3 // 4, None), slice(out.shape[-1] * 3 // 4, None))
    precision('fp32')
    ref = run(x * quadrant)
    precision('fp16x3')
    out = run(x * quadrant)
    if spec in (RANGE_CASES[0], RANGE_CASES[4]) and ref[sl].numel():
        err = float((out[sl] - ref[sl]).abs().max() / ref[sl].abs().max())
        assert err <= 1e-5, ('quiet corner', err)


def test_fp16_masked_dgrad_tiny_gradients(cuda, precision):
    """The masked data gradient (leaky-ReLU backward inside the 3x3 data-gradient kernel) on binary16 limbs with
    gradients of magnitude 1e-9: equal to mask pass + convolution in the exact-product fp32 kernels to 1e-5."""
    from gangealing_amd.op import conv_mfma as cm
    from gangealing_amd import _lib
    g = torch.Generator(device='cpu').manual_seed(21)
    for cin, cout, res, n in ((64, 96, 32, 2), (128, 128, 64, 8)):
        w = (torch.randn(cin, cout, 3, 3, generator=g) / 30).to(cuda)          # layer: cout -> cin channels
        pw = cm.PackedWeight(w, 1, cout, cin, 3, 1, 1, 1.0)                    # data-gradient pack (reduce over cin)
        dy = (torch.randn(n, cin, res, res, generator=g) * 1e-9).to(cuda)
        y = torch.randn(n, cin, res, res, generator=g).to(cuda)
        gm = torch.empty_like(dy)
        _lib.call('gg_fused_lrelu_bwd_f32', gm, None, dy, y, 0.2, 2 ** 0.5, n, cin, res * res)
        precision('fp32')
        ref = cm.conv_forward(gm, pw, n, 1, cin, cout, 3, 1, 1, 0)
        precision('fp16x3')
        assert cm.limb_code(grad=True) == 50          # binary16 limbs + the gradient-operand bit (round 6)
        dx = cm.masked_dgrad(dy, y, 0.2, 2 ** 0.5, pw, n, cin, cout, res, res)
        assert dx is not None
        err = float((dx - ref).abs().max() / ref.abs().max())
        assert err <= 1e-5, err


@pytest.mark.parametrize('mode_name,tol', [('bf16x3', 3e-5), ('fp16x3', 1e-5)])
@pytest.mark.parametrize('spec', [(2, 4, 32, 48, (33, 65)), (16, 16, 32, 48, (65, 65)), (1, 2, 64, 200, (9, 129))],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_s2_patch_grouped_vs_float64(spec, mode_name, tol, cuda, precision):
    """The stride-2 patch tile with groups (the reference's per-sample-weight formulation reaches it with groups = N:
    the data gradient of conv_transpose2d(groups = N), networks.py:268-272), forward and gradient launches, against
    float64 F.conv2d on the host.  The third case has one pixel tile per image: split-K."""
    import torch.nn.functional as F
    from gangealing_amd.op import conv_mfma as cm
    n, groups, cin_g, cout_g, (hh, ww) = spec
    g = torch.Generator(device='cpu').manual_seed(4321)
    x = torch.randn(n, groups * cin_g, hh, ww, generator=g)
    w = torch.randn(groups * cout_g, cin_g, 3, 3, generator=g) / (cin_g * 9) ** 0.5
    ref = F.conv2d(x.double(), w.double(), stride=2, groups=groups)
    pw = cm.PackedWeight(w.to(cuda), groups, cout_g, cin_g, 3, 0, 0, 1.0)
    precision(mode_name)
    for grad in (False, True):
        out = cm.conv_forward(x.to(cuda), pw, n, groups, cin_g, cout_g, 3, 2, 0, 0, grad=grad)
        assert out.shape == ref.shape
        err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
        assert err < tol, (grad, err)
        if cm._S2_PATCH and pw.split_ok():
            assert cm.last_conv_kernel().startswith('conv3x3s2_patch16'), cm.last_conv_kernel()


@pytest.mark.parametrize('mode_name,tol', [('bf16x3', 3e-5), ('bf16x6', 1e-5), ('fp16x3', 3e-5)])
@pytest.mark.parametrize('spec', [(2, 64, 128, 65, 65, 1), (3, 96, 64, 33, 129, 1), (1, 128, 160, 17, 33, 1),
                                  (2, 32, 64, 129, 129, 1), (4, 64, 96, 17, 41, 1), (2, 64, 96, 66, 130, 2),
                                  (16, 64, 128, 129, 129, 1), (5, 160, 136, 9, 65, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_s2_row_streaming_wgrad_vs_float64(spec, mode_name, tol, cuda, precision):
    """3x3 / stride 2 / pad 0 weight gradient (conv_s2_wgrad.hip: rolling window of column-parity planes) against the
    float64 gradient of F.conv2d on the host: 64- and 128-channel tiles, ragged cout / cin tiles, a 16- and a 20-column
    output (ragged strip), even input sizes, groups, chained K-units (batch 16)."""
    import torch.nn.functional as F
    from gangealing_amd.op import conv_mfma as cm
    n, cin, cout, h, w, groups = spec
    g = torch.Generator(device='cpu').manual_seed(77)
    x = torch.randn(n, cin * groups, h, w, generator=g)
    oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    dy = torch.randn(n, cout * groups, oh, ow, generator=g)
    wd = torch.zeros(cout * groups, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wd, stride=2, groups=groups).backward(dy.double())
    ref = 0.5 * wd.grad
    precision(mode_name)
    out = cm.conv_wgrad(x.to(cuda), dy.to(cuda), n, groups, cin, cout, 3, 2, 0, 0.5)
    assert out.shape == ref.shape
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < tol, err
    slot = torch.full_like(out, 2.0)
    cm.conv_wgrad(x.to(cuda), dy.to(cuda), n, groups, cin, cout, 3, 2, 0, 0.5, into=slot)
    err = float((slot.cpu().double() - 2.0 - ref).abs().max() / ref.abs().max())
    assert err < tol + 2e-6, err


@pytest.mark.parametrize('spec', [(2, 3, 128, 64, 64, True), (3, 1, 40, 64, 80, False), (16, 3, 128, 256, 256, True),
                                  (2, 4, 17, 128, 32, True)], ids=lambda s: 'x'.join(map(str, s)))
def test_conv1x1_few_input_channels(spec, cuda, precision):
    """1x1 convolution with <= 4 input channels (conv1x1_fewin_kernel: the data gradient of the last ToRGB layer, a
    write-only stream) against float64: style / demodulation scales, bias, ragged channel groups; every precision mode
    sends this shape to the same exact-fp32 kernel."""
    from gangealing_amd.op import conv_mfma as cm
    n, cin, cout, h, w, scaled = spec
    g = torch.Generator(device='cpu').manual_seed(31)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g)
    s_in = torch.rand(n, cin, generator=g) + 0.5 if scaled else None
    s_out = torch.rand(n, cout, generator=g) + 0.5 if scaled else None
    bias = torch.randn(cout, generator=g) if scaled else None
    ref = torch.einsum('nihw,oi->nohw', (x * s_in.view(n, cin, 1, 1) if scaled else x).double(), wt[:, :, 0, 0].double() * 0.7)
    if scaled:
        ref = ref * s_out.view(n, cout, 1, 1).double() + bias.view(1, cout, 1, 1).double()
    pw = cm.PackedWeight(wt.to(cuda), 1, cout, cin, 1, 0, 0, 0.7)
    for mode in ('fp32', 'fp16x3'):
        precision(mode)
        out = cm.conv_forward(x.to(cuda), pw, n, 1, cin, cout, 1, 1, 0, 0, in_scale=None if s_in is None else s_in.to(cuda),
                              out_scale=None if s_out is None else s_out.to(cuda), bias=None if bias is None else bias.to(cuda))
        assert cm.last_conv_kernel() == 'conv1x1_fewin', cm.last_conv_kernel()
        err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, err
