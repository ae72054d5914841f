"""The 1-bit sign plane of the fused "3x3 convolution + (noise) + bias + leaky ReLU" layers (round 5).

The leaky-ReLU backward uses the layer's saved OUTPUT only as a sign (reference models/stylegan2/op/fused_act.py:33-38 ->
fused_bias_act_kernel.cu:36-47).  The forward epilogue of the 3x3 patch tile writes that sign as one bit per element
(gg_modconv3x3_act_bits_f32) and the masked data gradient reads one word per pixel and 32 channels instead of 32 floats
(gg_conv3x3_masked_dgrad_bits_f32).  Index / bit work: everything here is compared bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def fp16x3():
    from gangealing_amd.op import conv_mfma
    old = conv_mfma.PRECISION
    conv_mfma.set_precision('fp16x3')
    yield conv_mfma
    conv_mfma.set_precision(old)


def pack_signs(y):
    """(N, C, H, W) -> uint32 (N, H*W, C/32): bit (c & 31) of word c / 32 = (y > 0)."""
    n, c, h, w = y.shape
    b = (y > 0).reshape(n, c // 32, 32, h * w).permute(0, 3, 1, 2).cpu().numpy().astype(np.uint64)
    return (b << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)


SHAPES = [
    # n, cin, cout, h, w, style-scaled      (>= 512 tiles each: launches with fewer are split along Cin, see below)
    (16, 128, 128, 64, 128, True),    # 256-pixel (8-wave, pipelined) tile
    (8, 96, 128, 64, 128, True),      # 128-pixel tile
    (16, 64, 64, 64, 64, False),      # 64-channel tiles (cout <= 64), the STN trunk's form
    (8, 128, 96, 64, 128, False),     # ragged co tile (96 = 3 x 32): the last 32-block of the 128-channel tile is empty
    (16, 128, 64, 64, 64, True),      # 64-channel tiles, style-scaled
]


@pytest.mark.parametrize('spec', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_sign_plane_and_bit_masked_gradient(spec, cuda, fp16x3):
    cm = fp16x3
    n, cin, cout, h, w, scaled = spec
    g = torch.Generator(device='cpu').manual_seed(17)
    x = torch.randn(n, cin, h, w, generator=g).to(cuda)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(cuda)
    style = (torch.rand(n, cin, generator=g) + 0.5).to(cuda) if scaled else None
    demod = (torch.rand(n, cout, generator=g) + 0.5).to(cuda) if scaled else None
    noise = torch.randn(n, 1, h, w, generator=g).to(cuda) if scaled else None
    nw = torch.tensor([0.3], device=cuda) if scaled else None
    bias = (torch.randn(cout, generator=g) * 0.1).to(cuda)
    fwd = cm.PackedWeight(wt, 1, cout, cin, 3, 0, 0, 1.0)
    bwd = cm.PackedWeight(wt, 1, cin, cout, 3, 1, 1, 1.0)
    act = (noise, nw, bias, 0.2, 2 ** 0.5)
    y0 = cm.conv_forward(x, fwd, n, 1, cin, cout, 3, 1, 1, 0, in_scale=style, out_scale=demod, act=act)
    y, bits = cm.conv_forward(x, fwd, n, 1, cin, cout, 3, 1, 1, 0, in_scale=style, out_scale=demod, act=act,
                              want_sign_bits=True)
    assert torch.equal(y, y0)                                  # the same launch, one more output
    assert bits is not None and bits.shape == (n, h * w, cout // 32) and bits.dtype == torch.int32
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), pack_signs(y))
    assert 0.2 < float((y > 0).float().mean()) < 0.8           # (both branches populated)
    dy = torch.randn(n, cout, h, w, generator=g).to(cuda) * 1e-4
    a = cm.masked_dgrad(dy, y, 0.2, 2 ** 0.5, bwd, n, cout, cin, h, w, demod, style)
    from gangealing_amd import _lib
    name_a = _lib.load().gg_last_conv_kernel().decode()
    b = cm.masked_dgrad(dy, None, 0.2, 2 ** 0.5, bwd, n, cout, cin, h, w, demod, style, sign_bits=bits)
    name_b = _lib.load().gg_last_conv_kernel().decode()
    assert a is not None and b is not None
    assert ',masked,' in name_a and ',bitmasked,' in name_b, (name_a, name_b)
    assert torch.equal(a, b)                                   # bitwise: the same mask, the same arithmetic


def test_split_k_launch_reports_no_plane(cuda, fp16x3):
    """A layer with too few tiles is split along Cin; its activation runs in the reduce pass, which writes no plane: the
    forward says so and the backward keeps the fp32 reference."""
    cm = fp16x3
    n, cin, cout, h, w = 1, 512, 128, 16, 16
    x = torch.randn(n, cin, h, w, device=cuda)
    wt = torch.randn(cout, cin, 3, 3, device=cuda) / (cin * 9) ** 0.5
    fwd = cm.PackedWeight(wt, 1, cout, cin, 3, 0, 0, 1.0)
    y, bits = cm.conv_forward(x, fwd, n, 1, cin, cout, 3, 1, 1, 0,
                              act=(None, None, torch.zeros(cout, device=cuda), 0.2, 1.0), want_sign_bits=True)
    assert bits is None and bool(torch.isfinite(y).all())


@pytest.mark.parametrize('node', ['stn-conv', 'styled-conv'])
def test_layer_gradients_equal_with_and_without_the_plane(node, cuda, fp16x3, monkeypatch):
    """The autograd nodes that own these layers (_Conv3x3BiasAct: STN trunk / VGG; _ModulatedConvAct: the generator's
    StyledConv) give bit-identical input gradients whether their backward masks from the plane or from the saved output."""
    cm = fp16x3
    g = torch.Generator(device='cpu').manual_seed(5)

    def run():
        if node == 'stn-conv':
            x = torch.randn(4, 64, 64, 64, generator=g).to(cuda).requires_grad_(True)
            wt = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(cuda)
            y = cm.conv3x3_bias_act(x, wt, torch.zeros(64, device=cuda) + 0.01)
        else:
            from gangealing_amd.stylegan2.networks import StyledConv
            torch.manual_seed(3)
            layer = StyledConv(128, 128, 3, 512).to(cuda).requires_grad_(False)
            x = torch.randn(2, 128, 32, 64, generator=g).to(cuda).requires_grad_(True)
            y = layer(x, torch.randn(2, 512, generator=g).to(cuda), noise=torch.randn(2, 1, 32, 64, generator=g).to(cuda))
        (gx,) = torch.autograd.grad(y, x, torch.randn(y.shape, generator=g).to(cuda))
        return y.detach(), gx

    g.manual_seed(5)
    y1, g1 = run()
    monkeypatch.setattr(cm, 'DISABLED', cm.DISABLED | {'sign_bits'})
    g.manual_seed(5)
    y2, g2 = run()
    assert torch.equal(y1, y2) and torch.equal(g1, g2)
