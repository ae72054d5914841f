"""Every convolution of the benchmark configuration (C2: generator 256^2, similarity+flow STN at 128^2, VGG16 at
128^2; per-GPU batch 16, i.e. 32 images through VGG) as an isolated operator at its exact shape, against what the
reference calls for it - F.conv2d / F.conv_transpose2d (conv2d_gradfix.py:34,66), here evaluated in float64 - forward,
data gradient and weight gradient, in the exact-fp32 and the bf16x3 arithmetic.

The batch decides which kernel variant is launched (256-pixel patch tiles, 128-wide transposed tiles, 64-channel
tiles, the row-streaming weight gradient, split-K fallbacks), so these are the kernels the benchmark runs.  A
convolution is linear: nothing can "flip" here (cf. tests/test_gpu_configs.py::check_grads), so the bounds are tight:
max |err| / max |ref| <= 2e-5 (fp32 kernels) and <= 5e-5 (bf16x3: two bf16 limbs per operand), both inside
north_star's 1e-4.  The modulated convolutions are compared with the reference's per-sample grouped formulation
(networks.py:243-280) including the fused noise + bias + leaky-ReLU tail, whose backward is checked with the
activation mask taken from the forward output (the sign reference the operator itself uses, fused_act.py:27-38).
"""
import math
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import record_parity

pytestmark = pytest.mark.gpu

MODES = ['fp32', 'bf16x3', 'fp16x3', 'bf16']
# 'bf16' (one bf16 limb per operand) is the plain-bf16 arithmetic of BASELINE.json's benchmark configuration, not a
# parity mode: operands carry 8 mantissa bits, errors of a 4608-term dot product are ~3e-3 of the largest output
# 'fp16x3': binary16 limbs with a per-tile block exponent on the forward AND (round 4) the data-gradient convolutions -
# fp32-class, held to the fp32 kernels' bound there; bf16 limbs on the weight gradients (the bf16x3 bound)
TOL = {'fp32': 2e-5, 'bf16x3': 5e-5, 'fp16x3': 5e-5, 'bf16': 2e-2}
FWD_TOL = {'fp16x3': 2e-5}                 # applies to 'forward' and 'dgrad' checks
N = 16


@pytest.fixture(params=MODES)
def mode(request):
    from gangealing_amd.op import conv_mfma
    old = conv_mfma.PRECISION
    conv_mfma.set_precision(request.param)
    yield request.param
    conv_mfma.set_precision(old)


def rnd(shape, seed, device, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(device)


def check(test, mode, name, got, ref):
    got, ref = got.detach().double(), ref.detach().double()
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    l2 = float((got - ref).norm() / ref.norm())
    from conftest import PARITY
    PARITY.setdefault('c2_layer_ops', {}).setdefault(mode, {})[f'{test}/{name}'] = dict(
        max_err_rel_to_max=err / scale, rel_l2_err=l2)
    tol = FWD_TOL.get(mode, TOL[mode]) if ('forward' in name or 'dgrad' in name) else TOL[mode]
    assert err <= tol * scale, (test, name, err / scale, l2)


# (name, batch, cin, cout, input size, kernel, stride, padding) - plain convolutions with trainable / frozen weights
PLAIN = [
    ('stn.stem 1x1 3->64 @128', N, 3, 64, 128, 1, 1, 0),
    ('stn.res 64->64 @128', N, 64, 64, 128, 3, 1, 1),
    ('stn.down 64->128 s2 (129->64)', N, 64, 128, 129, 3, 2, 0),
    ('stn.skip 1x1 64->128 s2 (127->64)', N, 64, 128, 127, 1, 2, 0),
    ('stn.res 128->128 @64', N, 128, 128, 64, 3, 1, 1),
    ('stn.down 128->512 s2 (65->32)', N, 128, 512, 65, 3, 2, 0),
    ('stn.skip 1x1 128->512 s2 (63->32)', N, 128, 512, 63, 1, 2, 0),
    ('stn.res 512->512 @32', N, 512, 512, 32, 3, 1, 1),
    ('stn.down 512->512 s2 (33->16)', N, 512, 512, 33, 3, 2, 0),
    ('stn.res 512->512 @16', N, 512, 512, 16, 3, 1, 1),
    ('stn.down 512->512 s2 (17->8)', N, 512, 512, 17, 3, 2, 0),
    ('stn.res 512->512 @8', N, 512, 512, 8, 3, 1, 1),
    ('stn.final 512->512 @4', N, 512, 512, 4, 3, 1, 1),
    ('flow.mask 512->576 @16', N, 512, 576, 16, 3, 1, 1),
    ('flow.out 512->2 @16', N, 512, 2, 16, 3, 1, 1),
    ('vgg 3->64 @128', 2 * N, 3, 64, 128, 3, 1, 1),
    ('vgg 64->64 @128', 2 * N, 64, 64, 128, 3, 1, 1),
    ('vgg 64->128 @64', 2 * N, 64, 128, 64, 3, 1, 1),
    ('vgg 128->128 @64', 2 * N, 128, 128, 64, 3, 1, 1),
    ('vgg 128->256 @32', 2 * N, 128, 256, 32, 3, 1, 1),
    ('vgg 256->256 @32', 2 * N, 256, 256, 32, 3, 1, 1),
    ('vgg 256->512 @16', 2 * N, 256, 512, 16, 3, 1, 1),
    ('vgg 512->512 @16', 2 * N, 512, 512, 16, 3, 1, 1),
    ('vgg 512->512 @8', 2 * N, 512, 512, 8, 3, 1, 1),
]


@pytest.mark.parametrize('spec', PLAIN, ids=lambda s: s[0])
def test_plain_conv_at_c2_shape(spec, mode, cuda):
    from gangealing_amd.op import conv_mfma
    name, n, cin, cout, size, k, stride, pad = spec
    seed = zlib.crc32(name.encode()) % 100000
    x = rnd((n, cin, size, size), seed, cuda).requires_grad_(True)
    w = rnd((cout, cin, k, k), seed + 1, cuda).requires_grad_(True)
    b = rnd((cout,), seed + 2, cuda, 0.1).requires_grad_(True)
    wscale = 1.0 / math.sqrt(cin * k * k)                      # EqualConv2d's runtime scale, folded into the pack
    y = conv_mfma.conv2d(x, w, b, stride=stride, padding=pad, weight_scale=wscale)
    dy = rnd(tuple(y.shape), seed + 3, cuda)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), dy)
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr * wscale, br, stride=stride, padding=pad)
    gxr, gwr, gbr = torch.autograd.grad(yr, (xr, wr, br), dy.double())
    check(name, mode, 'forward', y, yr)
    check(name, mode, 'dgrad', gx, gxr)
    check(name, mode, 'wgrad', gw, gwr)
    check(name, mode, 'bias grad', gb, gbr)


# (name, cin, cout, input size, upsample) - the generator's modulated 3x3 convolutions at batch 16
MODULATED = [
    ('g.conv1 512->512 @4', 512, 512, 4, False),
    ('g.up 512->512 4->8', 512, 512, 4, True),
    ('g.conv 512->512 @8', 512, 512, 8, False),
    ('g.up 512->512 8->16', 512, 512, 8, True),
    ('g.conv 512->512 @16', 512, 512, 16, False),
    ('g.up 512->512 16->32', 512, 512, 16, True),
    ('g.conv 512->512 @32', 512, 512, 32, False),
    ('g.up 512->512 32->64', 512, 512, 32, True),
    ('g.conv 512->512 @64', 512, 512, 64, False),
    ('g.up 512->256 64->128', 512, 256, 64, True),
    ('g.conv 256->256 @128', 256, 256, 128, False),
    ('g.up 256->128 128->256', 256, 128, 128, True),
    ('g.conv 128->128 @256', 128, 128, 256, False),
]


def reference_modulated(x, w, style, demod, upsample):
    """networks.py:243-280 per sample in float64: weight = scale * W * style[n, ci] (* demod[n, co]); transposed
    stride-2 convolution for the up-sampling layers (the Blur that follows is a separate operator)."""
    outs = []
    for i in range(x.shape[0]):
        wi = w * style[i].view(1, -1, 1, 1)
        if demod is not None:
            wi = wi * demod[i].view(-1, 1, 1, 1)
        if upsample:
            outs.append(F.conv_transpose2d(x[i:i + 1], wi.transpose(0, 1), stride=2, padding=0))
        else:
            outs.append(F.conv2d(x[i:i + 1], wi, padding=1))
    return torch.cat(outs, 0)


@pytest.mark.parametrize('spec', MODULATED, ids=lambda s: s[0])
def test_modulated_conv_at_c2_shape(spec, mode, cuda):
    from gangealing_amd.stylegan2.networks import ModulatedConv2d
    name, cin, cout, size, up = spec
    seed = zlib.crc32(name.encode()) % 100000
    mod = ModulatedConv2d(cin, cout, 3, 512, upsample=up).to(cuda)
    with torch.no_grad():
        mod.weight.copy_(rnd((1, cout, cin, 3, 3), seed, cuda))
        mod.modulation.weight.copy_(rnd((cin, 512), seed + 1, cuda))
        mod.modulation.bias.copy_(1.0 + rnd((cin,), seed + 2, cuda, 0.1))
    mod.requires_grad_(False)
    x = rnd((N, cin, size, size), seed + 3, cuda).requires_grad_(True)
    latent = rnd((N, 512), seed + 4, cuda).requires_grad_(True)
    # the module's own output includes the Blur for up-sampling layers; compare the convolution itself
    wmat_fwd, wmat_bwd, wsq = mod._weights()
    from gangealing_amd.op import conv_mfma
    style = mod.modulation(latent)
    y = conv_mfma.modulated_conv2d(x, style, wmat_fwd, wmat_bwd, wsq, 3, upsample=up, demodulate=True)
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
    check(name, mode, 'forward', y, yr)
    check(name, mode, 'dgrad', gx, gxr)
    check(name, mode, 'latent grad', glat, glr)


@pytest.mark.parametrize('spec', [s for s in MODULATED if not s[4]], ids=lambda s: s[0])
def test_fused_styled_conv_at_c2_shape(spec, mode, cuda):
    """StyledConv in one kernel (conv + noise + bias + leaky ReLU) and its masked data gradient (the frozen-generator
    path of the training step).  The backward reference uses the operator's own sign reference: the forward OUTPUT."""
    from gangealing_amd.stylegan2.networks import StyledConv
    name, cin, cout, size, _ = spec
    seed = zlib.crc32(name.encode()) % 100000 + 7
    layer = StyledConv(cin, cout, 3, 512).to(cuda)
    with torch.no_grad():
        layer.conv.weight.copy_(rnd((1, cout, cin, 3, 3), seed, cuda))
        layer.conv.modulation.weight.copy_(rnd((cin, 512), seed + 1, cuda))
        layer.conv.modulation.bias.copy_(1.0 + rnd((cin,), seed + 2, cuda, 0.1))
        layer.noise.weight.fill_(0.3)
        layer.activate.bias.copy_(rnd((cout,), seed + 3, cuda, 0.2))
    layer.requires_grad_(False)
    x = rnd((N, cin, size, size), seed + 4, cuda).requires_grad_(True)
    latent = rnd((N, 512), seed + 5, cuda)
    noise = rnd((N, 1, size, size), seed + 6, cuda)
    y = layer(x, latent, noise=noise)
    dy = rnd(tuple(y.shape), seed + 8, cuda)
    (gx,) = torch.autograd.grad(y, x, dy)
    conv = layer.conv
    xr = x.detach().double().requires_grad_(True)
    w64 = conv.weight[0].double() * conv.scale
    sr = F.linear(latent.double(), conv.modulation.weight.double() * conv.modulation.scale,
                  conv.modulation.bias.double() * conv.modulation.lr_mul)
    demod = torch.rsqrt((sr.pow(2) @ w64.pow(2).sum(dim=(2, 3)).t()) + 1e-8)
    pre = reference_modulated(xr, w64, sr, demod, False) + 0.3 * noise.double() + \
        layer.activate.bias.double().view(1, -1, 1, 1)
    yr = F.leaky_relu(pre, 0.2) * math.sqrt(2)
    # forward: compare away from the kink only through the value itself (continuous); backward: same mask on both sides
    check(name, mode, 'fused forward', y, yr)
    mask = torch.where(y.detach() > 0, 1.0, 0.2).double() * math.sqrt(2)
    (gxr,) = torch.autograd.grad(pre, xr, dy.double() * mask)
    check(name, mode, 'masked dgrad', gx, gxr)
