"""GPU tests of the 16-channel-chunk transposed 3x3 / stride-2 tile (csrc/conv_t_c16.hip, round 5): the operator the
generator's up-sampling ModulatedConv2d (reference models/stylegan2/networks.py:254-265: conv_transpose2d(stride 2))
and the data gradients of the STN's stride-2 convolutions run on.  Every case is forced through the new tile in both of
its forms (64 co x 128 q on four waves / 128 co on eight) with gg_set_tuning and compared with the exact-product fp32
MFMA kernel, with float64 conv_transpose2d, and with the round-4 tile on the same inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture
def precision():
    from gangealing_amd.op import conv_mfma
    old = conv_mfma.PRECISION
    yield conv_mfma.set_precision
    conv_mfma.set_precision(old)


@pytest.fixture
def tuning():
    from gangealing_amd import _lib

    def set_(mode, tw=64):
        _lib.set_tuning('GG_CONVT16', mode)
        _lib.set_tuning('GG_CONVT16_TW', tw)
    yield set_
    _lib.set_tuning('GG_CONVT16', _lib.TUNING_RESET)        # back to the environment's / built-in setting
    _lib.set_tuning('GG_CONVT16_TW', _lib.TUNING_RESET)


CASES = [
    # n, groups, cin_g, cout_g, (h, w), pad, in_scale
    (2, 1, 64, 130, (16, 16), 0, True),       # 16-wide tiles, ragged cout (three / two co tiles, the last ragged)
    (2, 1, 64, 128, (21, 32), 0, True),       # 32-wide tiles, ragged tile rows
    (1, 1, 96, 64, (33, 64), 1, False),       # pad 1 (data gradient of a stride-2 conv) + output_padding, 6 chunks
    (4, 1, 32, 128, (128, 128), 0, True),     # many tiles, two chunks
    (1, 1, 256, 256, (16, 16), 0, True),      # few tiles: split-K over the 16 chunks
    (2, 1, 64, 96, (4, 4), 0, True),          # 4-wide image: one tile covers the whole q-grid
    (3, 1, 64, 64, (7, 8), 1, False),         # 8-wide, pad 1
    (2, 1, 96, 80, (9, 16), 0, True),         # six chunks, ragged cout, 16-wide rows with a ragged last tile row
    (2, 3, 32, 40, (12, 32), 0, True),        # groups (the reference's per-sample weights: groups = N), ragged cout
    (1, 1, 32, 64, (130, 128), 0, False),     # two chunks; two 128 x 1 edge tiles per image
]


def _case(spec, cuda, seed=77):
    from gangealing_amd.op import conv_mfma as cm
    n, groups, cin, cout, (hh, ww), pad, scaled = spec
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(n, groups * cin, hh, ww, generator=g).to(cuda)
    # conv_transpose2d weight layout (in, out / groups, k, k); the packed operand is its (out, in) transposition
    w = (torch.randn(groups * cin, cout, 3, 3, generator=g) / (cin * 9) ** 0.5).to(cuda)
    s_in = (torch.rand(n, groups * cin, generator=g) + 0.5).to(cuda) if scaled else None
    s_out = (torch.rand(n, groups * cout, generator=g) + 0.5).to(cuda) if scaled else None
    bias = torch.randn(groups * cout, generator=g).to(cuda)
    pw = cm.PackedWeight(w, groups, cout, cin, 3, 1, 0, 0.7)
    out_hw = (2 * hh, 2 * ww) if pad == 1 else None
    run = lambda grad=False: cm.conv_forward(x, pw, n, groups, cin, cout, 3, 2, pad, 1, in_scale=s_in, out_scale=s_out,
                                             bias=bias, out_hw=out_hw, grad=grad)
    xs = x.double() * (s_in.double()[:, :, None, None] if scaled else 1.0)
    ref = F.conv_transpose2d(xs, w.double() * 0.7, stride=2, padding=pad, output_padding=1 if pad == 1 else 0,
                             groups=groups)
    if scaled:
        ref = ref * s_out.double()[:, :, None, None]
    ref = ref + bias.double()[None, :, None, None]
    return run, ref


@pytest.mark.parametrize('tw', [64, 16])
@pytest.mark.parametrize('tco', [64, 128])
@pytest.mark.parametrize('mode_name,tol', [('fp16x3', 4e-6), ('bf16x3', 3e-5)])
@pytest.mark.parametrize('spec', CASES, ids=lambda s: 'x'.join(map(str, s)))
def test_c16_tile_vs_float64(spec, mode_name, tol, tco, tw, cuda, precision, tuning):
    from gangealing_amd import _lib
    run, ref = _case(spec, cuda)
    precision(mode_name)
    tuning(tco, tw)
    for grad in (False, True):
        out = run(grad)
        name = _lib.load().gg_last_conv_kernel().decode()
        assert name.startswith(f'convT3x3s2_c16<limbs2,{tco}co'), name
        assert out.shape == ref.shape and bool(torch.isfinite(out).all())
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        assert err < tol, (grad, err)
    # the round-4 tile on the same launch: both are two-limb evaluations of the same sums
    tuning(0)
    old = run()
    assert 'c16' not in _lib.load().gg_last_conv_kernel().decode()
    assert float((old - out).abs().max() / ref.abs().max()) < 2 * tol


@pytest.mark.parametrize('tco', [64, 128])
@pytest.mark.parametrize('scale', [1e-30, 1e-9, 3e5, 1e30], ids=lambda v: f'{v:g}')
def test_c16_block_exponent_any_magnitude(scale, tco, cuda, precision, tuning):
    """binary16 limbs with the per-tile block exponent (conv_common.h: BlockExp) on the new tile: operands of any
    magnitude, exponent growth between 16-channel chunks, zero chunks."""
    from gangealing_amd.op import conv_mfma as cm
    n, cin, cout, hh, ww = 2, 96, 128, 24, 32
    g = torch.Generator(device='cpu').manual_seed(3)
    x = torch.randn(n, cin, hh, ww, generator=g).to(cuda) * scale
    steps = [1e-7, 1.0, 3e4, 0.0, 7e2, 1e-3]
    ramp = torch.tensor([steps[c // 16] for c in range(cin)], device=cuda).view(1, cin, 1, 1)
    w = (torch.randn(cin, cout, 3, 3, generator=g) / (cin * 9) ** 0.5).to(cuda)
    pw = cm.PackedWeight(w, 1, cout, cin, 3, 1, 0, 1.0)
    for xs in (x, x * ramp, x * ramp.flip(1)):
        precision('fp32')
        tuning(0)
        ref = cm.conv_forward(xs, pw, n, 1, cin, cout, 3, 2, 0, 1)
        assert bool(torch.isfinite(ref).all()) and float(ref.abs().max()) > 0
        precision('fp16x3')
        tuning(tco)
        for grad in (False, True):
            out = cm.conv_forward(xs, pw, n, 1, cin, cout, 3, 2, 0, 1, grad=grad)
            assert bool(torch.isfinite(out).all())
            err = float((out - ref).abs().max() / ref.abs().max())
            assert err <= 1e-5, (grad, err)


def test_c16_is_bitwise_repeatable(cuda, precision, tuning):
    run, _ = _case(CASES[4], cuda)          # the split-K case: partial copies added in a fixed order
    precision('fp16x3')
    tuning(64)
    a, b = run(), run()
    assert torch.equal(a, b)
