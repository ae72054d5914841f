"""Round 6 (second half): launches removed from the step's batch-independent tail.  Every fused route is held to the
route it replaces (the reference's operator order, still reachable through GG_DISABLE switches) on the same inputs:
forward, every parameter gradient and the input gradient.

  * ResBlock (models/stylegan2/networks.py:375-393 of the reference): 1 / sqrt(2) folded into the two branches, the skip
    branch's decimating blur as a tap node whose backward adds into conv1's data gradient.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def cm():
    from gangealing_amd.op import conv_mfma
    old, dis = conv_mfma.PRECISION, conv_mfma.DISABLED
    yield conv_mfma
    conv_mfma.set_precision(old)
    conv_mfma.DISABLED = dis


def _grads(block, x, g, cm, disabled):
    cm.DISABLED = frozenset(disabled)
    for p in block.parameters():
        p.grad = None
    xx = x.clone().requires_grad_(True)
    y = block(xx)
    y.backward(g)
    return y.detach(), xx.grad.detach(), {n: p.grad.detach().clone() for n, p in block.named_parameters()}


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3'])
@pytest.mark.parametrize('cin,cout,res,down', [(64, 128, 64, True), (128, 512, 32, True), (512, 512, 16, True),
                                               (512, 512, 8, True), (512, 512, 16, False), (64, 64, 33, True)])
def test_resblock_folded_equals_reference_order(cm, precision, cin, cout, res, down):
    """The folded block against the literal (conv2(conv1(x)) + skip(x)) / sqrt(2) on the same kernels: the scale moves
    from after the sum into the two addends, so the results agree to rounding (a few ulp of the block's output), and so
    do the gradients of every parameter and of the input."""
    from gangealing_amd.stylegan2.networks import ResBlock
    cm.set_precision(precision)
    torch.manual_seed(cin + res)
    dev = torch.device('cuda', 0)
    block = ResBlock(cin, cout, downsample=down).to(dev)
    with torch.no_grad():
        for n, p in block.named_parameters():
            if n.endswith('bias'):
                p.normal_(0.0, 0.2)
    n = 4
    x = torch.randn(n, cin, res, res, device=dev)
    out_res = block(x[:1]).shape[-1]
    g = torch.randn(n, cout, out_res, out_res, device=dev)
    base = set(cm.DISABLED)
    y0, dx0, gp0 = _grads(block, x, g, cm, base | {'resblock_fold'})
    if precision == 'fp32':
        # the exact-product mode keeps the reference's operation order (no fold): force the folded route for this test
        cm.DISABLED = frozenset(base - {'resblock_fold'})
        for p in block.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        y1 = block._forward_folded(xx)
        y1.backward(g)
        y1, dx1, gp1 = y1.detach(), xx.grad.detach(), {n_: p.grad.detach().clone() for n_, p in block.named_parameters()}
        assert torch.equal(block(x), y0)          # ... and block() itself is the reference order in this mode
    else:
        y1, dx1, gp1 = _grads(block, x, g, cm, base - {'resblock_fold'})
    assert y0.shape == y1.shape

    def close(a, b, tol):
        scale = float(b.abs().max()) + 1e-30
        return float((a - b).abs().max()) <= tol * scale

    assert close(y1, y0, 2e-6), float((y1 - y0).abs().max())
    assert close(dx1, dx0, 2e-5 if precision != 'fp32' else 4e-6), float((dx1 - dx0).abs().max())
    for name in gp0:
        assert close(gp1[name], gp0[name], 3e-5), (name, float((gp1[name] - gp0[name]).abs().max()))


def test_resblock_folded_route_is_taken_and_saves_launches(cm):
    """The folded block issues no separate scale pass in its backward: the merge's gradient is handed on as is, and the
    skip branch's adjoint blur accumulates into conv1's data gradient (gg_upfirdn2d_add_f32 instead of gg_upfirdn2d_f32 +
    an ATen add)."""
    from gangealing_amd import _lib
    from gangealing_amd.stylegan2.networks import ResBlock
    cm.set_precision('fp16x3')
    dev = torch.device('cuda', 0)
    block = ResBlock(128, 512).to(dev)
    x = torch.randn(2, 128, 32, 32, device=dev, requires_grad=True)
    seen = []
    orig = _lib.call

    def spy(name, *a, **k):
        seen.append(name)
        return orig(name, *a, **k)
    _lib.call = spy
    try:
        block(x).sum().backward()
    finally:
        _lib.call = orig
    assert 'gg_upfirdn2d_add_f32' in seen
    # the merge rides in the skip convolution, conv2's activation in its convolution: no element-wise pass of their own
    assert 'gg_conv1x1_split_residual_f32' in seen and 'gg_conv2d_split_act_f32' in seen
    assert 'gg_add_scale_f32' not in seen and 'gg_fused_bias_act_f32' not in seen


def test_blur_down_tap_matches_two_nodes():
    """(x, blur_down(x)) as one node: same forward as upfirdn2d(down=2), and the backward equals the sum autograd forms
    from the two separate paths - bitwise when only one path carries a gradient, to 1 ulp-of-sum otherwise."""
    from gangealing_amd.op.upfirdn2d import blur_down_tap, upfirdn2d
    from gangealing_amd.stylegan2.networks import make_kernel
    dev = torch.device('cuda', 0)
    k = make_kernel([1, 3, 3, 1]).to(dev)
    torch.manual_seed(3)
    for shape, pad in (((3, 8, 32, 32), (1, 1)), ((2, 5, 33, 33), (1, 1)), ((2, 4, 16, 16), (2, 1))):
        x = torch.randn(*shape, device=dev)
        xa = x.clone().requires_grad_(True)
        xp, xs = blur_down_tap(xa, k, pad)
        ref = upfirdn2d(x, k, down=2, pad=pad)
        assert torch.equal(xs, ref) and torch.equal(xp, x)
        g1, g2 = torch.randn_like(xp), torch.randn_like(xs)
        (xp * g1).sum().add((xs * g2).sum()).backward()
        xb = x.clone().requires_grad_(True)
        (xb * g1).sum().add((upfirdn2d(xb, k, down=2, pad=pad) * g2).sum()).backward()
        assert float((xa.grad - xb.grad).abs().max()) <= 4e-7 * float(xb.grad.abs().max())
        # only the decimated path
        xc = x.clone().requires_grad_(True)
        (blur_down_tap(xc, k, pad)[1] * g2).sum().backward()
        xd = x.clone().requires_grad_(True)
        (upfirdn2d(xd, k, down=2, pad=pad) * g2).sum().backward()
        assert torch.equal(xc.grad, xd.grad)


@pytest.mark.parametrize('n,heads', [(16, 1), (5, 4), (1, 1), (300, 3)])
def test_similarity_matrix_kernel_vs_reference_chain(n, heads):
    """gg_similarity_matrix_f32 against SimilarityHead.make_affine_matrix evaluated by ATen on the same device (the
    reference's 11-launch chain): the same operations in the same order - equal to the last bit, or within 1 ulp where the
    two builds' libm differ; the backward against autograd through that chain."""
    from gangealing_amd.spatial_transformers.flow_ops import similarity_matrix
    from gangealing_amd.spatial_transformers.warping_heads import SimilarityHead
    dev = torch.device('cuda', 0)
    torch.manual_seed(n + heads)
    p = (torch.randn(n, 4 * heads, device=dev) * 1.5)
    pa = p.clone().requires_grad_(True)
    pb = p.clone().requires_grad_(True)
    ma = similarity_matrix(pa, heads)
    mb = SimilarityHead.make_affine_matrix(*torch.split(pb, heads, dim=1))
    assert ma.shape == mb.shape == (n, heads, 2, 3)
    ulp = torch.finfo(torch.float32).eps * mb.abs().clamp_min(1e-30)
    assert bool(((ma - mb).abs() <= 2 * ulp).all()), float((ma - mb).abs().max())
    g = torch.randn_like(mb)
    ma.backward(g)
    mb.backward(g)
    assert float((pa.grad - pb.grad).abs().max()) <= 2e-6 * float(pb.grad.abs().max())


def test_similarity_head_uses_the_kernel(cm):
    from gangealing_amd import _lib
    from gangealing_amd.spatial_transformers.warping_heads import SimilarityHead
    dev = torch.device('cuda', 0)
    head = SimilarityHead(512, num_heads=1).to(dev)
    with torch.no_grad():
        head.linear.weight.normal_(0, 0.01)
    img = torch.rand(2, 3, 32, 32, device=dev)
    feats = torch.randn(2, 512, device=dev)
    seen, orig = [], _lib.call

    def spy(name, *a, **k):
        seen.append(name)
        return orig(name, *a, **k)
    _lib.call = spy
    try:
        out, grid, m, _ = head(img, feats)
    finally:
        _lib.call = orig
    assert 'gg_similarity_matrix_f32' in seen
    cm.DISABLED = frozenset(cm.DISABLED | {'similarity_matrix'})
    out2, grid2, m2, _ = head(img, feats)
    assert float((m.reshape(-1) - m2.reshape(-1)).abs().max()) <= 3e-7 * float(m2.abs().max())
    assert float((out - out2).abs().max()) <= 1e-5


def _unpack_bits(bits, cout):
    """(N, HW, cout / 32) int32 sign plane -> (N, cout, HW) bool."""
    n, hw, words = bits.shape
    b = bits.to(torch.int64) & 0xFFFFFFFF
    sh = torch.arange(32, device=bits.device, dtype=torch.int64)
    u = ((b.unsqueeze(-1) >> sh) & 1).reshape(n, hw, words * 32)[..., :cout]
    return u.permute(0, 2, 1).bool()


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3'])
@pytest.mark.parametrize('n,cin,cout,h,w,alpha,gain', [(4, 3, 64, 128, 128, 0.0, 1.0), (2, 3, 64, 36, 40, 0.2, 2 ** 0.5),
                                                       (3, 4, 32, 64, 64, 0.0, 1.0), (32, 3, 64, 128, 128, 0.0, 1.0),
                                                       (2, 1, 96, 128, 128, 0.2, 1.0)])
def test_few_input_channel_stem_forward_backward(cm, precision, n, cin, cout, h, w, alpha, gain):
    """The perceptual trunk's RGB stem (Conv2d(3, 64, 3, padding=1) + ReLU, lpips_backbones.py:109) on the streaming
    few-input-channel kernel: output against float64 conv2d + bias + (leaky) ReLU, the sign plane equal to pack(y > 0) bit
    for bit, and the input gradient through the sign-plane-masked few-output-channel kernel against float64 autograd."""
    import torch.nn.functional as F
    from gangealing_amd import _lib
    cm.set_precision(precision)
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(n * 7 + cin + h)
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(dev)
    b = (torch.randn(cout, generator=g) * 0.1).to(dev)
    dy = torch.randn(n, cout, h, w, generator=g).to(dev)
    seen, orig = [], _lib.call

    def spy(name, *a, **k):
        rc = orig(name, *a, **k)
        seen.append((name, cm.last_conv_kernel()))
        return rc
    xa = x.clone().requires_grad_(True)
    _lib.call = spy
    try:
        y = cm.conv3x3_bias_act(xa, wt, b, alpha, gain)
        y.backward(dy)
    finally:
        _lib.call = orig
    served = n * h * w >= 65536
    kernels = [k for _, k in seen]
    if served:
        assert any(k.startswith('conv3x3_fewin') for k in kernels), kernels
        assert any(name == 'gg_conv3x3_fewout_masked_bits_f32' for name, _ in seen), seen
    x64 = x.double().requires_grad_(True)
    z = F.conv2d(x64, wt.double(), b.double(), padding=1)
    y64 = F.leaky_relu(z, alpha) * gain
    assert float((y.double() - y64.detach()).abs().max()) <= 2e-6 * float(y64.abs().max())
    # gradient: a unit whose pre-activation lies within rounding distance of 0 may take the other branch in float32 (a
    # handful among the 33.5 M units of the largest case; each moves 9 * cin entries by O(|dy| |w|)), so the float64
    # reference is evaluated with the branch decisions the kernel stored (its output's sign - what its backward reads),
    # after checking that the two disagree only where the pre-activation is at rounding distance from the kink
    ours_pos = y.detach() > 0
    flips = ours_pos != (z.detach() > 0)
    assert int(flips.sum()) <= 64 and (not bool(flips.any()) or float(z.detach()[flips].abs().max()) <= 1e-5)
    (torch.where(ours_pos, z, z * alpha) * gain).backward(dy.double())
    d = xa.grad.double() - x64.grad
    assert float(d.abs().max()) <= 2e-5 * float(x64.grad.abs().max()), float(d.abs().max())


def test_few_input_channel_stem_sign_plane(cm):
    from gangealing_amd import _lib
    cm.set_precision('fp16x3')
    dev = torch.device('cuda', 0)
    torch.manual_seed(5)
    n, cin, cout, h, w = 8, 3, 64, 128, 128
    x = torch.randn(n, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.3
    b = torch.randn(cout, device=dev) * 0.1
    pw = cm.PackedWeight(wt, 1, cout, cin, 3, 0, 0, 1.0)
    y, bits = cm.conv_forward(x, pw, n, 1, cin, cout, 3, 1, 1, 0, act=(None, None, b, 0.0, 1.0), want_sign_bits=True)
    assert bits is not None and cm.last_conv_kernel().startswith('conv3x3_fewin')
    assert torch.equal(_unpack_bits(bits, cout), (y > 0).reshape(n, cout, h * w))
    # the masked few-output-channel gradient == the unmasked kernel on the explicitly masked gradient, bit for bit
    dy = torch.randn(n, cout, h, w, device=dev)
    wb = cm.PackedWeight(wt, 1, cin, cout, 3, 1, 1, 1.0)
    dx = torch.empty(n, cin, h, w, device=dev)
    _lib.call('gg_conv3x3_fewout_masked_bits_f32', dx, dy, bits, 0.2, 1.5, wb.fp32(), n, cout, cin, h, w)
    neg = torch.tensor(0.2, device=dev) * torch.tensor(1.5, device=dev)          # float32 product, as the kernel forms it
    masked = dy * torch.where(y > 0, torch.tensor(1.5, device=dev), neg)
    ref = cm.conv_forward(masked, wb, n, 1, cout, cin, 3, 1, 1, 0, grad=True)
    assert cm.last_conv_kernel().startswith('conv3x3_fewout')
    assert torch.equal(dx, ref)


@pytest.mark.parametrize('n,c,h,w', [(2, 8, 129, 129), (1, 3, 65, 65), (2, 4, 257, 257), (1, 5, 61, 93), (3, 2, 33, 33)])
def test_blur_sign_plane_forward_backward_bitwise(cm, n, c, h, w):
    """The up-sampling StyledConv's tail (blur + noise + bias + leaky ReLU, networks.py:268-298) with the activation's sign
    kept as a plane-major 1-bit plane: the plane equals (out > 0) bit for bit, the forward output is unchanged, and the
    adjoint blur that reads the plane reproduces the fp32-masked adjoint BITWISE."""
    from gangealing_amd import _lib
    from gangealing_amd.op.upfirdn2d import blur_noise_act
    from gangealing_amd.stylegan2.networks import make_kernel
    dev = torch.device('cuda', 0)
    torch.manual_seed(h + w)
    k = (make_kernel([1, 3, 3, 1]) * 4).to(dev)
    pad = (1, 1)
    x = torch.randn(n, c, h, w, device=dev)
    oh, ow = h + 2 - 3, w + 2 - 3
    noise = torch.randn(n, 1, oh, ow, device=dev)
    nw = torch.tensor([0.3], device=dev)
    b = torch.randn(c, device=dev) * 0.2
    g = torch.randn(n, c, oh, ow, device=dev)
    res = {}
    for route in ('fp32', 'bits'):
        cm.DISABLED = frozenset(cm.DISABLED | {'blur_bits'}) if route == 'fp32' else frozenset(cm.DISABLED - {'blur_bits'})
        xa = x.clone().requires_grad_(True)
        seen, orig = [], _lib.call

        def spy(name, *a, **kw):
            seen.append(name)
            return orig(name, *a, **kw)
        _lib.call = spy
        try:
            y = blur_noise_act(xa, k, pad, noise, nw, b, 0.2, 2 ** 0.5)
            y.backward(g)
        finally:
            _lib.call = orig
        res[route] = (y.detach(), xa.grad.detach(), seen)
    assert res['bits'][2].count('gg_blur4_fused_bits_f32') == 2 and 'gg_blur4_fused_bits_f32' not in res['fp32'][2]
    assert torch.equal(res['bits'][0], res['fp32'][0])
    assert torch.equal(res['bits'][1], res['fp32'][1])
    # the plane itself
    from gangealing_amd.op.upfirdn2d import blur_bits_words, blur_bits_unpack
    lib = _lib.load()
    assert lib.gg_blur4_bits_words(oh, ow) == blur_bits_words(oh, ow)
    out = torch.empty(n, c, oh, ow, device=dev)
    bits = torch.full((n * c, blur_bits_words(oh, ow)), -1, dtype=torch.int32, device=dev)   # stale contents must not survive
    _lib.call('gg_blur4_fused_bits_f32', out, x, k, n, c, h, w, 1, 1, 1, 1, noise, nw, b, bits, 0.2, 2 ** 0.5)
    assert torch.equal(out, res['fp32'][0])
    assert torch.equal(blur_bits_unpack(bits, oh, ow), (out > 0).reshape(n * c, oh, ow))
    # every word is written (columns / rows beyond the plane as zeros): nothing of the -1 fill survives
    words = (bits.to(torch.int64) & 0xFFFFFFFF).reshape(n * c, -1, 2)
    assert int((words[..., 1] >> 29).max()) == 0        # bits 61 .. 63 of each 64-bit cell row


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3'])
def test_flow_head_conv_relu_fused_equals_module_sequence(cm, precision):
    """FlowHead's conv3x3 -> ReLU -> conv3x3 stacks (warping_heads.py:160-169) with the first convolution's bias + ReLU in
    its epilogue: same low-resolution flow / mask logits and the same parameter gradients as the nn.Sequential route."""
    from gangealing_amd.spatial_transformers.warping_heads import FlowHead
    cm.set_precision(precision)
    dev = torch.device('cuda', 0)
    torch.manual_seed(11)
    head = FlowHead((1, 512, 16, 16)).to(dev)
    with torch.no_grad():
        for seq in (head.flow_out, head.mask_out):
            seq[2].weight.normal_(0, 0.02)
            seq[0].bias.normal_(0, 0.1)
    feats = torch.randn(4, 512, 16, 16, device=dev)
    base = set(cm.DISABLED)
    res = {}
    for route in ('seq', 'fused'):
        cm.DISABLED = frozenset(base | {'head_relu'}) if route == 'seq' else frozenset(base - {'head_relu'})
        for p in head.parameters():
            p.grad = None
        f = feats.clone().requires_grad_(True)
        low, mask = head._head(head.flow_out, f), head._head(head.mask_out, f)
        (low.square().sum() + mask.square().sum() * 0.01).backward()
        res[route] = (low.detach(), mask.detach(), f.grad.detach(), {n: p.grad.clone() for n, p in head.named_parameters()
                                                                     if p.grad is not None})
    tol = 2e-6 if precision == 'fp32' else 2e-5
    for a, b in zip(res['fused'][:3], res['seq'][:3]):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max())
    assert set(res['fused'][3]) == set(res['seq'][3])
    for name, gb in res['seq'][3].items():
        ga = res['fused'][3][name]
        assert float((ga - gb).abs().max()) <= 5 * tol * float(gb.abs().max()) + 1e-12, name



@pytest.mark.parametrize('shape,pad', [((16, 64, 17, 17), (1, 1)), ((4, 32, 9, 9), (2, 1)), ((2, 8, 19, 23), (1, 1)),
                                       ((3, 5, 5, 5), (2, 2)), ((2, 16, 16, 16), (2, 1))])
def test_small_plane_blur_constant_trip_count_path(shape, pad):
    """4x4 FIR with up = down = 1 on planes below the streaming kernel's 24 x 24 minimum (the generator's and the STN's
    blurs at <= 16^2) on upfirdn2d_direct's constant-trip-count path: against float64 correlation with the flipped taps,
    forward and adjoint (the adjoint is the same operator with flipped taps and the adjoint padding)."""
    import torch.nn.functional as F
    from gangealing_amd.op.upfirdn2d import upfirdn2d
    from gangealing_amd.stylegan2.networks import make_kernel
    dev = torch.device('cuda', 0)
    torch.manual_seed(shape[2] * 7 + shape[3])
    k = make_kernel([1, 3, 3, 1]).to(dev)
    x = torch.randn(*shape, device=dev)
    xa = x.clone().requires_grad_(True)
    y = upfirdn2d(xa, k, pad=pad)
    g = torch.randn_like(y)
    y.backward(g)
    x64 = x.double().requires_grad_(True)
    n, c, h, w = shape
    xp = F.pad(x64, (pad[0], pad[1], pad[0], pad[1]))
    w64 = torch.flip(k.double(), [0, 1])[None, None].expand(c, 1, 4, 4)
    y64 = F.conv2d(xp, w64, groups=c)
    y64.backward(g.double())
    assert y.shape == y64.shape
    assert float((y.double() - y64).abs().max()) <= 1e-6 * float(y64.abs().max())
    assert float((xa.grad.double() - x64.grad).abs().max()) <= 1e-6 * float(x64.grad.abs().max())


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3'])
@pytest.mark.parametrize('n,cin,cout,res', [(4, 64, 64, 128), (2, 128, 128, 64), (2, 512, 512, 32), (2, 32, 64, 40),
                                            (2, 512, 512, 16)])
def test_conv_act_blur_node_equals_the_three_separate_nodes(cm, precision, n, cin, cout, res):
    """ResBlock's conv1 -> FusedLeakyReLU -> Blur as one node whose backward applies the activation's backward in the
    Blur's adjoint (gg_blur4_act_bwd_f32): forward, input gradient and weight gradient BITWISE those of the three
    separate nodes (the masked gradient is the same tensor: explicit FMA chains in the blur, the same mask expression);
    the bias gradient to rounding (another, equally fixed summation order).  16^2 planes take the node's fallback."""
    from gangealing_amd import _lib
    from gangealing_amd.op.upfirdn2d import upfirdn2d
    from gangealing_amd.stylegan2.networks import make_kernel
    cm.set_precision(precision)
    dev = torch.device('cuda', 0)
    torch.manual_seed(cin + res)
    k = make_kernel([1, 3, 3, 1]).to(dev)
    pad = (2, 2)
    x = torch.randn(n, cin, res, res, device=dev)
    wt = (torch.randn(cout, cin, 3, 3, device=dev))
    b = torch.randn(cout, device=dev) * 0.2
    wscale = 1.0 / math.sqrt(cin * 9)
    gshape = (n, cout, res + 1, res + 1)
    g = torch.randn(*gshape, device=dev)
    res_ = {}
    for route in ('separate', 'node'):
        xa, wa, ba = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
        seen, orig = [], _lib.call

        def spy(name, *a, **kw):
            seen.append(name)
            return orig(name, *a, **kw)
        _lib.call = spy
        try:
            if route == 'separate':
                y = upfirdn2d(cm.conv3x3_bias_act(xa, wa, ba, 0.2, 2 ** 0.5, weight_scale=wscale), k, pad=pad)
            else:
                y = cm.conv3x3_bias_act_blur(xa, wa, ba, k, pad, 0.2, 2 ** 0.5, weight_scale=wscale)
            assert tuple(y.shape) == gshape
            y.backward(g)
        finally:
            _lib.call = orig
        res_[route] = (y.detach(), xa.grad, wa.grad, ba.grad, seen)
    if res >= 24:
        assert 'gg_blur4_act_bwd_f32' in res_['node'][4] and not any('lrelu_bwd' in s for s in res_['node'][4])
    assert torch.equal(res_['node'][0], res_['separate'][0])
    assert torch.equal(res_['node'][1], res_['separate'][1])
    assert torch.equal(res_['node'][2], res_['separate'][2])
    db0, db1 = res_['separate'][3], res_['node'][3]
    assert float((db1 - db0).abs().max()) <= 2e-5 * float(db0.abs().max())


@pytest.mark.parametrize('precision', ['fp32', 'fp16x3'])
@pytest.mark.parametrize('n,cin,cout,res,k,stride,pad', [
    (16, 512, 512, 4, 3, 1, 1),      # the similarity trunk's final 3x3 at 4^2
    (16, 512, 512, 9, 3, 2, 0),      # ... its last down-sampling convolution: 9^2 -> 4^2
    (16, 512, 512, 4, 1, 1, 0),      # ... and that block's 1x1 skip
    (3, 40, 72, 5, 3, 1, 1),         # channel counts off the tile sizes, 25 positions
    (2, 8, 33, 3, 1, 1, 0),
    (16, 512, 2, 16, 3, 1, 1),       # the flow head's last layer (few output channels)
    (5, 96, 3, 20, 3, 1, 1),
    (4, 17, 1, 8, 3, 1, 1)])
def test_small_layer_weight_gradients(cm, precision, n, cin, cout, res, k, stride, pad):
    """Weight gradients of the layers the K-slab kernels do not serve - outputs of <= 32 positions per image
    (wgrad_tiny_spatial_kernel) and <= 4 output channels (wgrad3x3_fewout_kernel): against float64, into a fresh tensor
    and accumulating into a slot, and through the entry point the training path uses."""
    from gangealing_amd import _lib
    cm.set_precision(precision)
    dev = torch.device('cuda', 0)
    torch.manual_seed(cin * 3 + res + k)
    x = torch.randn(n, cin, res, res, device=dev)
    oh = (res + 2 * pad - k) // stride + 1
    dy = torch.randn(n, cout, oh, oh, device=dev)
    base = torch.randn(cout, cin, k, k, device=dev)
    fresh = cm.conv_wgrad(x, dy, n, 1, cin, cout, k, stride, pad, 0.37)
    kernel = cm.last_conv_kernel()
    into = base.clone()
    cm.conv_wgrad(x, dy, n, 1, cin, cout, k, stride, pad, 0.37, into=into)
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, k, k), dy.double(), stride=stride, padding=pad) * 0.37
    scale = float(ref.abs().max())
    assert float((fresh.double() - ref).abs().max()) <= 2e-6 * scale, kernel
    assert float((into.double() - (base.double() + ref)).abs().max()) <= 2e-6 * (scale + float(base.abs().max())), kernel
    if cout <= 4 and k == 3:
        assert kernel == 'wgrad3x3_fewout', kernel
    elif k == 1 and oh * oh <= 32 and (oh * oh) % 32:
        assert kernel.startswith('wgrad_tiny_spatial'), kernel
