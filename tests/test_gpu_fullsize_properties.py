"""GPU tests at BASELINE.json's FULL C2 sizes (batch 16, 256^2 generator tensors), where the CPU oracle
would take minutes: size-independent properties instead of element-wise comparison -
adjointness <A x, g> == <x, A^T g>, linearity, identity warps, reduction consistency."""
import pytest
import torch

pytestmark = pytest.mark.gpu

N = 16


def rel(a, b):
    a, b = float(a.detach()), float(b.detach())
    return abs(a - b) / max(abs(a), abs(b), 1e-12)


def dot(a, b):
    return (a.double() * b.double()).sum()


def adjoint_error(y, g, x, gx):
    """|<y,g> - <x,gx>| normalised by ||y|| ||g|| (the inner products themselves nearly cancel for random data)."""
    num = abs(float(dot(y, g).detach()) - float(dot(x, gx).detach()))
    return num / (float(y.detach().double().norm()) * float(g.double().norm()) + 1e-30)


def test_upfirdn2d_blur_adjoint_full_size(cuda):
    from gangealing_amd.op import upfirdn2d
    k = torch.tensor([1., 3., 3., 1.], device=cuda)
    k = (k[None] * k[:, None]) / 64 * 4
    x = torch.randn(N, 128, 257, 257, device=cuda, requires_grad=True)      # largest blur of the path: 541 MB in
    y = upfirdn2d(x, k, pad=(1, 1))
    assert y.shape == (N, 128, 256, 256)
    g = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, x, g)
    assert adjoint_error(y, g, x, gx) < 1e-6
    # a constant image stays constant away from the border (taps sum to the x4 gain)
    c = upfirdn2d(torch.ones(1, 1, 257, 257, device=cuda), k, pad=(1, 1))
    torch.testing.assert_close(c[..., 2:-2, 2:-2], torch.full_like(c[..., 2:-2, 2:-2], 4.0), atol=1e-5, rtol=0)


def test_fused_lrelu_full_size_reductions(cuda):
    from gangealing_amd.op import fused_leaky_relu
    x = torch.randn(N, 128, 256, 256, device=cuda, requires_grad=True)       # 537 MB
    b = torch.randn(128, device=cuda, requires_grad=True)
    y = fused_leaky_relu(x, b)
    g = torch.randn_like(y)
    gx, gb = torch.autograd.grad(y, (x, b), g)
    # positive homogeneity of lrelu: y == where(y > 0, 1, 0.2) * sqrt(2) * (x + b)
    slope = torch.where(y > 0, 1.0, 0.2) * 2 ** 0.5
    torch.testing.assert_close(y, slope * (x + b.view(1, -1, 1, 1)), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(gx, slope * g, atol=1e-5, rtol=1e-5)
    ref_gb = gx.double().sum(dim=(0, 2, 3))
    assert float(((gb.double() - ref_gb).abs() / (ref_gb.abs() + 1.0)).max()) < 1e-3     # fp32 sum of 1e6 terms


@pytest.mark.parametrize('precision_mode', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('shape', [(128, 128, 256, 3, 1, 1, False), (256, 128, 128, 3, 2, 0, True),
                                   (512, 512, 64, 3, 1, 1, False)], ids=['conv256', 'upconv128to257', 'conv64'])
def test_modulated_conv_adjoint_full_size(shape, precision_mode, cuda):
    """<conv(x), g> == <x, dgrad(g)> for the generator's heaviest layers at batch 16 (the forward and the
    dgrad run different packed weights / kernels, so this checks them against each other)."""
    from gangealing_amd.op import conv_mfma as cm
    cin, cout, h, k, stride, pad, up = shape
    old = cm.PRECISION
    cm.set_precision(precision_mode)
    try:
        w = torch.randn(cout, cin, k, k, device=cuda) / (cin * k * k) ** 0.5
        fwd = cm.PackedWeight(w, 1, cout, cin, k, 0, 0)
        bwd = cm.PackedWeight(w, 1, cin, cout, k, 1, 0 if up else 1)
        wsq = w.pow(2).sum(dim=(2, 3))
        x = torch.randn(N, cin, h, h, device=cuda, requires_grad=True)
        style = (torch.rand(N, cin, device=cuda) + 0.5)
        y = cm.modulated_conv2d(x, style, fwd, bwd, wsq, k, upsample=up, demodulate=True)
        assert y.shape[-1] == (2 * h + 1 if up else h)
        g = torch.randn_like(y)
        (gx,) = torch.autograd.grad(y, x, g)
        assert adjoint_error(y, g, x, gx) < (1e-6 if precision_mode == 'fp32' else 2e-5)
        # linearity in x
        y2 = cm.modulated_conv2d(2.0 * x.detach(), style, fwd, bwd, wsq, k, upsample=up, demodulate=True)
        assert float((y2 - 2 * y).abs().max() / y.abs().max()) < 1e-5
    finally:
        cm.set_precision(old)


def test_stn_conv_weight_gradient_consistency_full_size(cuda):
    """<conv_w(x), g> == <w, wgrad(x, g)> on the STN's 512->512 @32^2 layer at batch 16, both arithmetic modes."""
    from gangealing_amd.op import conv_mfma as cm
    old = cm.PRECISION
    try:
        for mode, tol in (('fp32', 1e-6), ('bf16x3', 2e-5)):
            cm.set_precision(mode)
            x = torch.randn(N, 512, 32, 32, device=cuda)
            w = (torch.randn(512, 512, 3, 3, device=cuda) / 68).requires_grad_(True)
            y = cm.conv2d(x, w, padding=1, weight_scale=0.5)
            g = torch.randn_like(y)
            (gw,) = torch.autograd.grad(y, w, g)
            assert adjoint_error(y, g, w, gw) < tol
    finally:
        cm.set_precision(old)


def test_mipmap_warp_identity_and_flip_full_size(cuda):
    from gangealing_amd.spatial_transformers.antialiased_sampling import MipmapWarp
    from gangealing_amd.spatial_transformers.flow_ops import affine_grid
    img = torch.randn(N, 3, 256, 256, device=cuda)
    eye = torch.tensor([[1., 0, 0], [0, 1, 0]], device=cuda).repeat(N, 1, 1)
    warp = MipmapWarp(3.5).to(cuda)
    out = warp(img, affine_grid(eye, (N, 3, 256, 256)), padding_mode='reflection')
    # identity grid at native resolution; fp32 rounding of the coordinates (~3e-5 px) times pixel differences
    torch.testing.assert_close(out, img, atol=5e-4, rtol=0)
    assert float(warp.levels_map.max()) == 0.0
    flip = torch.tensor([[-1., 0, 0], [0, 1, 0]], device=cuda).repeat(N, 1, 1)
    out = warp(img, affine_grid(flip, (N, 3, 256, 256)), padding_mode='border')
    torch.testing.assert_close(out, img.flip(3), atol=5e-4, rtol=0)
    # 2x zoom-out onto a 128 grid: every pixel at mip level ~1 (coordinates step 255/127 px)
    out = warp(img, affine_grid(eye, (N, 3, 128, 128)), padding_mode='border')
    lv = warp.levels_map * 2.5
    assert 0.95 < float(lv.min()) <= float(lv.max()) < 1.05 and out.shape == (N, 3, 128, 128)


def test_adam_ema_full_arena(cuda):
    from gangealing_amd import _lib
    n = 43054278                                                               # STN sim+flow parameter count
    p = torch.randn(n, device=cuda)
    g = torch.randn(n, device=cuda)
    m, v, ema = torch.zeros_like(p), torch.zeros_like(p), p.clone()
    p0 = p.clone()
    _lib.call('gg_adam_ema_f32', p, m, v, ema, g, n, 1e-3, 0.9, 0.999, 1e-8, 1, 0.99, 1.0)
    big = g.abs() > 1e-3                                                        # |g| >> eps
    torch.testing.assert_close((p - p0)[big], -1e-3 * torch.sign(g)[big], atol=2e-6, rtol=1e-3)   # first step = -lr*sign(g)
    torch.testing.assert_close(ema, 0.99 * p0 + 0.01 * p, atol=1e-6, rtol=1e-5)
