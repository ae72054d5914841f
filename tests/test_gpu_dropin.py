"""The literal drop-in route on the GPU.  An unmodified reference networks.py on these operators
(`python -m gangealing_amd.launch .../train.py`) issues, per modulated layer, the call sequence of
models/stylegan2/networks.py:243-282: per-sample weights scale * W * style (* demod) materialised as an
(N * Cout, Cin, 3, 3) tensor, then conv2d_gradfix.conv2d(groups=N) - or conv_transpose2d(groups=N) followed by Blur for
the up-sampling layers.  gangealing_amd.stylegan2.networks.modconv_form('grouped') runs exactly that sequence through
gangealing_amd.op.conv2d_gradfix; here every modulated 3x3 layer of the benchmark configuration (batch 16: groups = 16,
up to 512 channels, 151 MB of per-sample weights per layer) is compared with a float64 evaluation of the same
formulation - forward, data gradient and the latent gradient, which in this form flows through the grouped
convolution's WEIGHT gradient - and the whole generator in this form is compared with the reference's golden image."""
import zlib

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from test_gpu_c2_layer_ops import MODULATED, N, check, reference_modulated, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['fp32', 'bf16x3', 'fp16x3'])
def mode(request):
    from gangealing_amd.op import conv_mfma
    old = conv_mfma.PRECISION
    conv_mfma.set_precision(request.param)
    yield request.param
    conv_mfma.set_precision(old)


@pytest.mark.parametrize('spec', MODULATED, ids=lambda s: s[0])
def test_grouped_modconv_at_c2_shape(spec, mode, cuda):
    from gangealing_amd.stylegan2 import networks
    name, cin, cout, size, up = spec
    seed = zlib.crc32(name.encode()) % 100000 + 31
    mod = networks.ModulatedConv2d(cin, cout, 3, 512, upsample=up).to(cuda)
    with torch.no_grad():
        mod.weight.copy_(rnd((1, cout, cin, 3, 3), seed, cuda))
        mod.modulation.weight.copy_(rnd((cin, 512), seed + 1, cuda))
        mod.modulation.bias.copy_(1.0 + rnd((cin,), seed + 2, cuda, 0.1))
    mod.requires_grad_(False)
    x = rnd((N, cin, size, size), seed + 3, cuda).requires_grad_(True)
    latent = rnd((N, 512), seed + 4, cuda).requires_grad_(True)
    calls = []
    from gangealing_amd.op import conv2d_gradfix
    real = (conv2d_gradfix.conv2d, conv2d_gradfix.conv_transpose2d)

    def spy(fn, kind):
        def wrapped(input, weight, *a, **kw):
            calls.append((kind, tuple(input.shape), tuple(weight.shape), kw.get('groups')))
            return fn(input, weight, *a, **kw)
        return wrapped
    conv2d_gradfix.conv2d, conv2d_gradfix.conv_transpose2d = spy(real[0], 'conv2d'), spy(real[1], 'conv_transpose2d')
    try:
        with networks.modconv_form('grouped'):
            blur, mod.blur = (mod.blur, torch.nn.Identity()) if up else (None, None)   # compare the convolution itself
            y = mod(x, latent)
            if up:
                mod.blur = blur
    finally:
        conv2d_gradfix.conv2d, conv2d_gradfix.conv_transpose2d = real
    # the reference's call: one grouped convolution on (1, N*Cin, H, W) with the materialised per-sample weights
    if up:
        assert calls == [('conv_transpose2d', (1, N * cin, size, size), (N * cin, cout, 3, 3), N)], calls
    else:
        assert calls == [('conv2d', (1, N * cin, size, size), (N * cout, cin, 3, 3), N)], calls
    dy = rnd(tuple(y.shape), seed + 5, cuda)
    gx, glat = torch.autograd.grad(y, (x, latent), dy)
    xr = x.detach().double().requires_grad_(True)
    lr = latent.detach().double().requires_grad_(True)
    w64 = mod.weight[0].double() * mod.scale
    sr = F.linear(lr, mod.modulation.weight.double() * mod.modulation.scale,
                  mod.modulation.bias.double() * mod.modulation.lr_mul)
    demod = torch.rsqrt((sr.pow(2) @ w64.pow(2).sum(dim=(2, 3)).t()) + 1e-8)
    yr = reference_modulated(xr, w64, sr, demod, up)
    gxr, glr = torch.autograd.grad(yr, (xr, lr), dy.double())
    check('dropin ' + name, mode, 'forward', y, yr)
    check('dropin ' + name, mode, 'dgrad', gx, gxr)
    check('dropin ' + name, mode, 'latent grad (through the grouped weight gradient)', glat, glr)


def test_generator_in_reference_form_matches_the_reference(mode, cuda):
    """Generator(256), batch 16, every modulated layer in the grouped form: the reference's golden image."""
    from gangealing_amd.stylegan2 import Generator, networks
    from test_gpu_configs import load_det, D, check_batch
    (c,) = load_golden('c2_generator')
    n = c['meta']['batch']
    g = load_det(Generator(256, 512, 8)).to(cuda).eval().requires_grad_(False)
    noise = [D(f'c2gen.noise{i}', (n, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)), cuda) for i in range(g.num_layers)]
    with torch.no_grad(), networks.modconv_form('grouped'):
        img, _ = g([torch.from_numpy(c['z']).to(cuda)], noise=noise)
    check_batch('dropin_generator', mode, img, c, 'img')
    # and with a gradient to the latent (the first layers' per-sample weights then need their weight gradient)
    w = torch.from_numpy(c['w']).to(cuda).requires_grad_(True)
    with networks.modconv_form('grouped'):
        img2, _ = g([w.unsqueeze(1).repeat(1, g.n_latent, 1)], input_is_latent=True, noise=noise)
    check_batch('dropin_generator', mode, img2, c, 'img_from_w')
    from test_gpu_configs import check_one_grad
    img2.backward(D('c2gen.gimg', tuple(img2.shape), cuda))
    check_one_grad('dropin_generator', mode, 'gw', w.grad.cpu().numpy(), c['gw'], c['gw64'])
