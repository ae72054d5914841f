"""Round 6: the up-sampling convolution's operand written once in MFMA-ready form (csrc/conv_t_c16.hip PRELIMB loader,
gg_modconv3x3_act_amax_f32 -> gg_torgb_limb_f32 -> gg_convT3x3s2_prelimb_f32; include/gangealing_hip.h).

Checked: the fused ToRGB pass reproduces the stand-alone ToRGB convolution BITWISE; the limb-form consumer reproduces the
fp32-operand kernel BITWISE while the per-image exponent is 0 (the same limbs reach the matrix pipe) and stays within the
binary16-limb bound against float64 for operands of any magnitude (per-image instead of per-tile exponent); the epilogue's
per-image maxima are exact; a whole Generator(256) forward + backward agrees with the route on and off to 2e-6 of the
image scale, and the route is really taken; shapes the limb-form tile does not serve fall back."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture
def cm():
    from gangealing_amd.op import conv_mfma
    old, dis = conv_mfma.PRECISION, conv_mfma.DISABLED
    conv_mfma.set_precision('fp16x3')
    yield conv_mfma
    conv_mfma.set_precision(old)
    conv_mfma.DISABLED = dis


def styled_layer(cm, n, cin, cout, res, scale, seed, cuda):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, cin, res, res, generator=g) * scale).to(cuda)
    style = (torch.randn(n, cin, generator=g) * 0.3 + 1.0).to(cuda)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    wsq = (w.pow(2).sum(dim=(2, 3))).to(cuda)
    demod = torch.rsqrt((style * style) @ wsq.t() + 1e-8)
    pw = cm.PackedWeight(w.to(cuda), 1, cout, cin, 3, 0, 0, 1.0)
    noise = torch.randn(n, 1, res, res, generator=g).to(cuda)
    nw = torch.tensor([0.1]).to(cuda)
    ab = (torch.randn(cout, generator=g) * 0.1).to(cuda)
    return x, style, demod, pw, (noise, nw, ab, 0.2, 2 ** 0.5)


@pytest.mark.parametrize('scale', [1.0, 1e-6, 3e4], ids=['O(1)', '1e-6', '3e4'])
def test_limb_route_against_fp32_operand_route(scale, cuda, cm):
    n, c, c_up, res = 16, 128, 64, 64            # (512 tiles of 128 pixels: no split-K, the tile's own epilogue writes y)
    x, style, demod, pw, act = styled_layer(cm, n, c, c, res, 1.0, 3, cuda)
    g = torch.Generator().manual_seed(4)
    amax = torch.zeros(n, device=cuda)
    y, bits = cm.conv_forward(x, pw, n, 1, c, c, 3, 1, 1, 0, in_scale=style, out_scale=demod, act=act,
                              want_sign_bits=True, amax_out=amax)
    assert cm.last_amax_written() and bits is not None
    assert torch.equal(amax, y.abs().amax(dim=(1, 2, 3)))                      # per-image maxima, exact
    y_plain, bits_plain = cm.conv_forward(x, pw, n, 1, c, c, 3, 1, 1, 0, in_scale=style, out_scale=demod, act=act,
                                          want_sign_bits=True)
    assert torch.equal(y, y_plain) and torch.equal(bits, bits_plain)           # the amax epilogue changes nothing else
    y = y * scale
    amax = amax * scale
    # ToRGB + limb form
    w_rgb = (torch.randn(3, c, 1, 1, generator=g) / c ** 0.5).to(cuda)
    rgb_pw = cm.PackedWeight(w_rgb, 1, 3, c, 1, 0, 0, 1.0)
    s_rgb = (torch.randn(n, c, generator=g) * 0.3 + 1.0).to(cuda)
    b_rgb = torch.randn(3, generator=g).to(cuda)
    s_up = (torch.randn(n, c, generator=g) * 0.3 + 1.0).to(cuda)
    res_l = cm.torgb_limb(y, rgb_pw, s_rgb, b_rgb, s_up, amax)
    assert res_l is not None
    rgb, xlimb, xexp = res_l
    rgb_ref = cm.conv_forward(y, rgb_pw, n, 1, c, 3, 1, 1, 0, 0, in_scale=s_rgb, bias=b_rgb)
    assert torch.equal(rgb, rgb_ref)
    bound = (amax * s_up.abs().amax(dim=1)).cpu().numpy()
    e = xexp.cpu().numpy()
    for b, ee in zip(bound, e):
        assert (ee == 0) == (0.125 <= b <= 2048.0), (b, ee)
        if ee != 0:
            assert 64.0 <= b * 2.0 ** -ee < 128.0, (b, ee)
    # the up-sampling convolution on it
    w_up = torch.randn(c, c_up, 3, 3, generator=g) / (3 * c ** 0.5)
    up_pw = cm.PackedWeight(w_up.to(cuda), 1, c_up, c, 3, 1, 0, 1.0)
    d_up = (torch.rand(n, c_up, generator=g) + 0.5).to(cuda)
    out_l = cm.conv_forward(y, up_pw, n, 1, c, c_up, 3, 2, 0, 1, in_scale=s_up, out_scale=d_up, prelimb=(xlimb, xexp))
    assert cm.last_conv_kernel().endswith('prelimb>'), cm.last_conv_kernel()
    out_f = cm.conv_forward(y, up_pw, n, 1, c, c_up, 3, 2, 0, 1, in_scale=s_up, out_scale=d_up)
    assert not cm.last_conv_kernel().endswith('prelimb>')
    if scale == 1.0:
        assert (e == 0).all() and torch.equal(out_l, out_f)
    ref = F.conv_transpose2d((y * s_up[:, :, None, None]).double().cpu(), w_up.double(), stride=2) * \
        d_up.double().cpu()[:, :, None, None]
    for out in (out_l, out_f):
        err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err <= 1e-5, err


def test_shapes_off_the_limb_tile_fall_back(cuda, cm):
    """A 12-pixel-wide input (not a power of two) is not served by the 16-channel-chunk tile: GG_NOT_SERVED, nothing
    launched, and conv_forward takes the fp32 operand - the result is the plain route's."""
    n, c, c_up, hh, ww = 2, 64, 64, 12, 12
    g = torch.Generator().manual_seed(9)
    y = torch.randn(n, c, hh, ww, generator=g).to(cuda)
    s_up = (torch.randn(n, c, generator=g) * 0.3 + 1.0).to(cuda)
    w_up = (torch.randn(c, c_up, 3, 3, generator=g) / 24).to(cuda)
    up_pw = cm.PackedWeight(w_up, 1, c_up, c, 3, 1, 0, 1.0)
    bogus = (torch.zeros(n * c * hh * ww * 2, dtype=torch.int16, device=cuda), torch.zeros(n, dtype=torch.int32, device=cuda))
    a = cm.conv_forward(y, up_pw, n, 1, c, c_up, 3, 2, 0, 1, in_scale=s_up, prelimb=bogus)
    b = cm.conv_forward(y, up_pw, n, 1, c, c_up, 3, 2, 0, 1, in_scale=s_up)
    assert torch.equal(a, b) and float(a.abs().max()) > 0


def test_generator_is_bitwise_the_same_with_and_without_the_limb_route(cuda, cm):
    from gangealing_amd.stylegan2 import Generator
    from test_gpu_configs import load_det, D
    n = 4
    gen = load_det(Generator(256, 512, 8)).to(cuda).eval().requires_grad_(False)
    noise = [D(f'prelimb.noise{i}', (n, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)), cuda) for i in range(gen.num_layers)]
    z = D('prelimb.z', (n, 512), cuda)
    seen = []
    real = cm.torgb_limb

    def spy(*a, **k):
        r = real(*a, **k)
        seen.append(None if r is None else tuple(a[0].shape))
        return r
    out = {}
    for route in ('on', 'off'):
        cm.DISABLED = frozenset(cm.DISABLED | {'prelimb'}) if route == 'off' else frozenset(cm.DISABLED - {'prelimb'})
        cm.torgb_limb = spy
        try:
            with torch.no_grad():                                        # generator pass 1: no autograd node at all
                img0, lat = gen([z], return_latents=True, noise=noise)
            w = lat[:, 0].detach().clone().requires_grad_(True)          # pass 2: gradient w.r.t. the first 5 W+ slots
            latent = w.unsqueeze(1).repeat(1, gen.n_latent, 1)
            img1, _ = gen([latent], input_is_latent=True, noise=noise, grad_latents=5)
            (gw,) = torch.autograd.grad(img1, w, D('prelimb.g', tuple(img1.shape), cuda))
        finally:
            cm.torgb_limb = real
        out[route] = (img0, img1, gw, list(seen))
        seen.clear()
    # taken at input resolutions 64 and 128 of both passes (at batch 4 the 32^2 layer runs split-K: its reduce pass, not the
    # tile's epilogue, writes y, so no per-image maximum exists and that layer keeps the fp32 route; at batch 16 all three)
    taken = [s[-1] for s in out['on'][3] if s is not None]
    assert taken in ([64, 128, 64, 128], [32, 64, 128, 32, 64, 128]), out['on'][3]
    assert out['off'][3] == []
    # Equal to the accuracy of the arithmetic, not bitwise: the limb route takes ONE exponent per image from the bound
    # max |y| * max |style| where the fp32-operand tile takes one per tile and chunk from the operand itself - with the
    # deterministic test weights that bound leaves the E = 0 band on some layers, so the same values are split at another
    # scale (both splits keep 22 bits)
    for a, b, tol in zip(out['on'][:2], out['off'][:2], (2e-6, 2e-6)):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()), float((a - b).abs().max() / b.abs().max())
    # The latent gradient is bounded as a DECISION-dependent quantity: a leaky-ReLU unit whose pre-activation the two
    # splits round to different sides of 0 takes the other branch in one route and moves single entries by 5e-5 ... 2e-4 of
    # the largest (which units those are follows the last ulp of everything upstream: 2e-5 / 5e-5 / 1.8e-4 were measured
    # on three builds of round 6 that differ only in the rounding of a blur or a bias sum).  The arithmetic of the two
    # routes is compared where nothing can flip: the images above, and layer by layer in the tests above (bitwise at E = 0).
    ga, gb = out['on'][2], out['off'][2]
    assert float((ga - gb).norm()) <= 5e-4 * float(gb.norm()), float((ga - gb).norm() / gb.norm())
    assert float((ga - gb).abs().max()) <= 2e-3 * float(gb.abs().max()), float((ga - gb).abs().max() / gb.abs().max())
    assert float(ga.abs().max()) > 0
