"""GPU parity tests (run with -m gpu on the MI355X box): every HIP operator, called through the
C ABI via the drop-in Python modules, against (a) the golden vectors produced by the reference's
own CPU bodies and (b) the numpy/torch-CPU oracle on seeded inputs.
Tolerances: fp32 activations 1e-4 (north_star); integer by-products bit-exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-4


def T(a, dev, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.requires_grad_(True) if grad else t


def close(t, ref, atol=TOL, rtol=1e-4):
    np.testing.assert_allclose(t.detach().cpu().numpy(), ref, atol=atol, rtol=rtol)


@pytest.mark.parametrize('case', load_golden('upfirdn2d'), ids=lambda c: c['meta']['tag'])
def test_upfirdn2d_golden(case, cuda):
    from gangealing_amd.op import upfirdn2d
    m = case['meta']
    pad = m['pad']
    x = T(case['x'], cuda, True)
    k = T(case['k'], cuda)
    if pad[0] == pad[2] and pad[1] == pad[3]:
        out = upfirdn2d(x, k, up=m['up'], down=m['down'], pad=(pad[0], pad[1]))
    else:   # asymmetric x/y pads are reachable only below the python signature (as in the reference)
        from gangealing_amd.op.upfirdn2d import UpFirDn2d
        out = UpFirDn2d.apply(x, k, (m['up'],) * 2, (m['down'],) * 2, tuple(pad))
    close(out, case['out'], 1e-5)
    out.backward(T(case['g'], cuda))
    close(x.grad, case['gx'], 1e-5)


@pytest.mark.parametrize('case', load_golden('half_ops'), ids=lambda c: c['meta']['kind'])
def test_half_tensors_against_reference(case, cuda):
    """binary16 upfirdn2d / fused_leaky_relu (the reference dispatches half: upfirdn2d_kernel.cu:311,
    fused_bias_act_kernel.cu:89) against the reference's CPU bodies run on half tensors.  The HIP kernels compute in
    fp32 and round once: within 1.5 (upfirdn2d) / 2.5 (three roundings in the reference's activation) half ulps of the
    reference's result, and correctly rounded with respect to the fp32 evaluation of the same inputs."""
    import torch
    from gangealing_amd.op import upfirdn2d, fused_leaky_relu
    m = case['meta']
    ulp = 2.0 ** -10

    def near(a, b, ulps=1.5):
        a, b = a.float().cpu(), torch.from_numpy(np.asarray(b)).float()
        tol = ulps * ulp * b.abs().clamp_min(float(b.abs().max()) * 2 ** -4)
        assert a.dtype == torch.float32 and bool(((a - b).abs() <= tol).all()), float(((a - b).abs() / tol).max())

    x = torch.from_numpy(case['x']).to(cuda).requires_grad_(True)
    assert x.dtype == torch.float16
    g = torch.from_numpy(case['g']).to(cuda)
    if m['kind'] == 'upfirdn2d':
        pad = m['pad']
        out = upfirdn2d(x, torch.from_numpy(case['k']).to(cuda), up=m['up'], down=m['down'], pad=(pad[0], pad[1]))
        assert out.dtype == torch.float16
        near(out, case['out'])
        exact = torch.from_numpy(case['out32'])          # fp32 evaluation of the same rounded inputs
        assert bool(((out.float().cpu() - exact).abs() <= 2.0 ** -11 * exact.abs() + 2e-6 * float(exact.abs().max())).all())
        out.backward(g)
        near(x.grad, case['gx'])
    else:
        b = torch.from_numpy(case['b']).to(cuda).requires_grad_(True)
        out = fused_leaky_relu(x, b, m['negative_slope'], m['scale'])
        assert out.dtype == torch.float16
        near(out, case['out'], ulps=2.5)            # the reference rounds after the add, the slope and the gain
        out.backward(g)
        near(x.grad, case['gx'], ulps=2.5)
        assert b.grad.dtype == torch.float16
        near(b.grad, case['gb'], ulps=4.0)          # a sum of up to 192 rounded terms, accumulated in fp32 here


def test_upfirdn2d_hot_shapes_vs_oracle(cuda):
    from gangealing_amd.op import upfirdn2d
    from oracle import np_ops
    rs = np.random.RandomState(1)
    k = (np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0).astype(np.float32)
    for (shape, gain, pad) in [((2, 4, 129, 129), 4, (1, 1)), ((1, 3, 257, 257), 4, (1, 1)),
                               ((2, 2, 128, 128), 1, (2, 2)), ((2, 2, 128, 128), 1, (1, 1)),
                               ((1, 2, 200, 75), 1, (2, 2)), ((1, 1, 65, 65), 4, (1, 1)),
                               # full-width row-band kernel: odd last column, partial last band, 64-wide groups
                               ((1, 2, 256, 256), 1, (2, 2)), ((2, 3, 64, 64), 1, (2, 2)), ((1, 2, 50, 131), 1, (1, 1)),
                               ((1, 2, 37, 66), 4, (1, 1))]:
        x = rs.randn(*shape).astype(np.float32)
        out = upfirdn2d(T(x, cuda), T(k * gain, cuda), pad=pad)
        close(out, np_ops.upfirdn2d(x, k * gain, pad=(pad[0], pad[1], pad[0], pad[1])), 1e-5)


def test_upfirdn2d_resampling_paths_vs_oracle(cuda):
    """The generic kernel's constant-trip-count paths (4x4 taps, up 2 / down 2) and its plane loop: odd and non-square
    planes, the ToRGB / ResBlock-skip pads, more planes than the grid's y extent covers in one pass."""
    from gangealing_amd.op import upfirdn2d
    from oracle import np_ops
    rs = np.random.RandomState(3)
    k = (np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0).astype(np.float32)
    for (shape, up, down, pad) in [((2, 3, 16, 16), 2, 1, (2, 1)), ((1, 2, 17, 13), 2, 1, (2, 1)), ((1, 1, 1, 1), 2, 1, (2, 1)),
                                   ((2, 5, 32, 32), 1, 2, (1, 1)), ((1, 3, 33, 21), 1, 2, (1, 1)), ((1, 2, 15, 64), 1, 2, (2, 2)),
                                   ((1, 2, 16, 16), 1, 2, (0, 0)), ((3, 700, 8, 8), 1, 2, (1, 1)), ((3, 700, 4, 4), 2, 1, (2, 1)),
                                   ((1, 2, 9, 11), 2, 2, (2, 1)),
                                   # row-walk kernel of the 4x4 / down-2 FIR (upfirdn2d_fir4_down2_kernel): even and odd
                                   # left pads, pads beyond the taps' reach, a C2-sized plane set, tiny planes
                                   ((2, 5, 64, 64), 1, 2, (2, 1)), ((1, 3, 31, 130), 1, 2, (3, 2)), ((4, 64, 128, 128), 1, 2, (1, 1)),
                                   ((1, 2, 7, 9), 1, 2, (1, 1)), ((3, 9, 20, 12), 1, 2, (4, 4)), ((2, 130, 32, 32), 1, 2, (1, 1))]:
        x = rs.randn(*shape).astype(np.float32)
        kk = k * (up * up)
        out = upfirdn2d(T(x, cuda), T(kk, cuda), up=up, down=down, pad=pad)
        close(out, np_ops.upfirdn2d(x, kk, up=(up, up), down=(down, down), pad=(pad[0], pad[1], pad[0], pad[1])), 1e-5)


def test_upfirdn2d_f64_gradcheck(cuda):
    from gangealing_amd.op import upfirdn2d
    k = torch.rand(4, 4, dtype=torch.float64, device=cuda)
    for up, down, pad in [(1, 1, (1, 1)), (2, 1, (2, 1)), (1, 2, (1, 1))]:
        x = torch.randn(1, 2, 7, 6, dtype=torch.float64, device=cuda, requires_grad=True)
        assert torch.autograd.gradcheck(lambda t: upfirdn2d(t, k, up=up, down=down, pad=pad), (x,), eps=1e-6, atol=1e-6)
        assert torch.autograd.gradgradcheck(lambda t: upfirdn2d(t, k, up=up, down=down, pad=pad), (x,), eps=1e-6, atol=1e-6)


@pytest.mark.parametrize('case', load_golden('fused_act'))
def test_fused_act_golden(case, cuda):
    from gangealing_amd.op import fused_leaky_relu
    x, b = T(case['x'], cuda, True), T(case['b'], cuda, True)
    out = fused_leaky_relu(x, b)
    close(out, case['out'], 1e-6)
    out.backward(T(case['g'], cuda))
    nz = (case['x'] + case['b'].reshape([1, -1] + [1] * (case['x'].ndim - 2))) != 0
    np.testing.assert_allclose(x.grad.cpu().numpy()[nz], case['gx'][nz], atol=1e-6)
    if nz.all():
        close(b.grad, case['gb'], 1e-5)


def test_fused_act_module_and_large(cuda):
    from gangealing_amd.op import FusedLeakyReLU
    from oracle import np_ops
    rs = np.random.RandomState(2)
    for shape in [(3, 64, 33, 31), (2, 128, 64, 64), (16, 512)]:
        x = rs.randn(*shape).astype(np.float32)
        mod = FusedLeakyReLU(shape[1]).to(cuda)
        with torch.no_grad():
            mod.bias.copy_(T(rs.randn(shape[1]).astype(np.float32), cuda))
        xt = T(x, cuda, True)
        out = mod(xt)
        ref = np_ops.fused_leaky_relu(x, mod.bias.detach().cpu().numpy())
        close(out, ref, 1e-6)
        g = rs.randn(*shape).astype(np.float32)
        out.backward(T(g, cuda))
        gx, gb = np_ops.fused_leaky_relu_backward(g, ref)
        close(xt.grad, gx, 1e-6)
        close(mod.bias.grad, gb, 2e-3, 1e-4)      # fp32 sum of up to 2*64*64*... terms, atomics order


def test_fused_act_f64_gradcheck(cuda):
    from gangealing_amd.op import fused_leaky_relu
    x = torch.randn(2, 3, 4, 5, dtype=torch.float64, device=cuda, requires_grad=True)
    b = torch.randn(3, dtype=torch.float64, device=cuda, requires_grad=True)
    assert torch.autograd.gradcheck(fused_leaky_relu, (x, b), eps=1e-6, atol=1e-6)
    assert torch.autograd.gradgradcheck(fused_leaky_relu, (x, b), eps=1e-6, atol=1e-6)


@pytest.mark.parametrize('case', load_golden('grid_sample'), ids=lambda c: c['meta']['padding_mode'] + '-' + c['meta']['grid'])
def test_warp_golden(case, cuda):
    from gangealing_amd.spatial_transformers.antialiased_sampling import Warp
    out = Warp()(T(case['x'], cuda), T(case['grid'], cuda), padding_mode=case['meta']['padding_mode'])
    close(out, case['out'], 1e-5)


@pytest.mark.parametrize('case', load_golden('mipmap_warp') + load_golden('mipmap_warp_deep'),
                         ids=lambda c: f"{c['x'].shape[-1]}-{c['meta']['padding_mode']}-{c['meta']['grid']}")
def test_mipmap_warp_golden(case, cuda):
    from gangealing_amd.spatial_transformers.antialiased_sampling import MipmapWarp
    from oracle import np_ops
    m = case['meta']
    x, grid = T(case['x'], cuda, True), T(case['grid'], cuda, True)
    # (mipmap_warp_deep: the reference's default constructor - 8 levels - and 4.5; antialiased_sampling.py:22-60)
    warp = (MipmapWarp() if m['grid'] == 'default_ctor_identity' else MipmapWarp(max_num_levels=m['max_num_levels'])).to(cuda)
    assert warp.max_num_levels == m['max_num_levels']
    out = warp(x, grid, padding_mode=m['padding_mode'])
    close(out, case['out'], 2e-5)
    close(warp.levels_map, case['levels_map'], 1e-6)
    out.backward(T(case['g'], cuda))
    close(grid.grad, case['ggrid'], 2e-4, 2e-4)
    close(x.grad, case['gx'], 2e-5)
    # integer by-products of the same sampling, from the kernel's own index output, on every pixel (no masking):
    # floor / ceil of the level equal floor / ceil of the REFERENCE's level map stored in the fixture
    from gangealing_amd.spatial_transformers.antialiased_sampling import warp_indices
    h_in, w_in = case['x'].shape[2], case['x'].shape[3]
    _, _, lo, hi = warp_indices(grid.detach(), h_in, w_in, m['max_num_levels'], 0.0, m['padding_mode'])
    ref_lv = case['levels_map'].astype(np.float32) * np.float32(m['max_num_levels'] - 1.0)
    exact = np.abs(ref_lv - np.round(ref_lv)) > 1e-6      # levels_map was divided by 2.5 and re-multiplied: 1-ulp noise
    assert np.array_equal(lo.cpu().numpy()[exact], np.floor(ref_lv)[exact].astype(np.int32))
    assert np.array_equal(hi.cpu().numpy()[exact], np.ceil(ref_lv)[exact].astype(np.int32))
    # (tests/test_gpu_indices.py compares against the un-rescaled reference levels with no exclusions at all)


def test_sampling_indices_non_power_of_two_image(cuda):
    """floor(ix), floor(iy) on a non-square, non-power-of-two image for all three padding modes, every point, against
    ATen's formulas restated in numpy float32 (oracle/np_ops.grid_source_coords); the reference-pinned special-point
    cases live in tests/test_gpu_indices.py."""
    from gangealing_amd.spatial_transformers.antialiased_sampling import warp_indices
    from oracle import np_ops
    rs = np.random.RandomState(5)
    h, w = 37, 53
    grid = (rs.rand(2, 40, 40, 2).astype(np.float32) * 3 - 1.5)
    for mode in ['border', 'reflection', 'zeros']:
        gx_, gy_, _, _ = warp_indices(T(grid, cuda), h, w, padding_mode=mode, antialias=False)
        ix, iy = np_ops.grid_source_coords(grid, h, w, mode)
        assert np.array_equal(gx_.cpu().numpy(), np.floor(ix).astype(np.int32))
        assert np.array_equal(gy_.cpu().numpy(), np.floor(iy).astype(np.int32))


@pytest.mark.parametrize('case', load_golden('bilinear_downsample'))
def test_bilinear_downsample_golden(case, cuda):
    from gangealing_amd.spatial_transformers.antialiased_sampling import BilinearDownsample
    x = T(case['x'], cuda, True)
    out = BilinearDownsample(case['meta']['stride'], 3).to(cuda)(x)
    close(out, case['out'], 1e-5)
    out.backward(T(case['g'], cuda))
    close(x.grad, case['gx'], 1e-5)


def test_similarity_grid_golden(cuda):
    from gangealing_amd.spatial_transformers.flow_ops import affine_grid
    (c,) = load_golden('similarity_head')
    theta = T(c['composed'], cuda, True)
    grid = affine_grid(theta, (3, 3, 16, 16))
    close(grid, c['grid'], 1e-5)
    # gradient of the grid wrt theta against torch's own affine_grid on the CPU
    g = torch.from_numpy(c['g'])
    tc = torch.from_numpy(c['composed']).requires_grad_(True)
    torch.nn.functional.affine_grid(tc, (3, 3, 16, 16), align_corners=False).backward(g)
    grid.backward(g.to(cuda))
    close(theta.grad, tc.grad.numpy(), 1e-4)


@pytest.mark.parametrize('case', load_golden('flow_head'))
def test_flow_compose_golden(case, cuda):
    from gangealing_amd.spatial_transformers.flow_ops import flow_compose, flow_resize
    low = T(np.ascontiguousarray(case['low'].transpose(0, 3, 1, 2)), cuda, True)     # (N,h,w,2) -> NCHW
    mask, base = T(case['mask'], cuda, True), T(case['base'], cuda, True)
    flow, delta = flow_compose(low, mask, base, 8)
    close(delta, case['delta'], 1e-5)
    close(flow, case['flow'], 1e-5)
    torch.autograd.backward([flow, delta], [T(case['g_flow'], cuda), T(case['g_delta'], cuda)])
    close(low.grad, case['glow'].transpose(0, 3, 1, 2), 1e-4)
    close(mask.grad, case['gmask'], 1e-4)
    close(base.grad, case['gbase'], 2e-3, 1e-4)
    close(flow_resize(T(case['flow'], cuda), 2.0), case['resized2x'], 1e-5)


def test_flow_resize_grad(cuda):
    from gangealing_amd.spatial_transformers.flow_ops import flow_resize
    f = torch.randn(2, 6, 6, 2, device=cuda, requires_grad=True)
    out = flow_resize(f, 2.0)
    g = torch.randn_like(out)
    out.backward(g)
    fc = f.detach().cpu().requires_grad_(True)
    ref = torch.nn.functional.interpolate(fc.permute(0, 3, 1, 2), scale_factor=2.0, mode='bilinear').permute(0, 2, 3, 1)
    ref.backward(g.cpu())
    close(out, ref.detach().numpy(), 1e-5)
    close(f.grad, fc.grad.numpy(), 1e-5)


@pytest.mark.parametrize('case', load_golden('flow_losses'))
def test_flow_losses_golden(case, cuda):
    from gangealing_amd.spatial_transformers.flow_ops import flow_losses
    d = T(case['delta'], cuda, True)
    losses = flow_losses(d)
    close(losses[0], case['tv'], 1e-6, 1e-5)
    close(losses[1], case['identity'], 1e-6, 1e-5)
    (case['meta']['tv_weight'] * losses[0] + case['meta']['id_weight'] * losses[1]).backward()
    close(d.grad, case['gdelta'], 1e-5, 1e-4)


@pytest.mark.parametrize('case', load_golden('splat2d'))
def test_splat2d_golden(case, cuda):
    """tests/golden/splat2d.npz holds outputs of the reference's own kernel (meta.source == 'reference-kernel':
    utils/splat2d_cuda/src/splat_gpu_impl.cu compiled unmodified, oracle/Makefile + oracle/make_golden_splat.py)."""
    from gangealing_amd.splat2d_cuda import splat2d
    assert case['meta']['source'] == 'reference-kernel'
    out = splat2d(T(case['input'], cuda), T(case['coords'], cuda), T(case['values'], cuda), T(case['sigma'], cuda),
                  case['meta']['soft_normalize'])
    # both sides accumulate with float atomics in arbitrary order: compare relative to the plane's largest value
    ref = case['out']
    err = float(np.abs(out.cpu().numpy() - ref).max())
    assert err <= 2e-5 * max(1.0, float(np.abs(ref).max())), err


@pytest.mark.parametrize('shape', [(2, 3, 256, 256, 4096, 1.3), (1, 1, 512, 512, 20000, 3.0), (3, 4, 37, 53, 300, 0.6)])
def test_splat2d_against_live_reference_kernel(shape, cuda):
    """When the compiled reference kernel travelled to this box (oracle/_ref/libsplat_ref.so), run both on larger
    random inputs than the fixtures hold."""
    from oracle import make_golden_splat as ref
    if not ref.reference_available():
        pytest.skip('oracle/_ref/libsplat_ref.so not present')
    from gangealing_amd.splat2d_cuda import splat2d
    n, c, h, w, p, sig = shape
    g = torch.Generator().manual_seed(p)
    coords = (torch.rand(n, p, 2, generator=g) * torch.tensor([w + 8.0, h + 8.0]) - 4.0).to(cuda)
    values = torch.randn(n, p, c, generator=g).to(cuda)
    sigma = torch.full((n,), sig, device=cuda)
    inp = torch.randn(n, c, h, w, generator=g).to(cuda)
    for soft in (False, True):
        want = ref.reference_splat2d(inp, coords, values, sigma, soft)
        got = splat2d(inp, coords, values, sigma, soft)
        err = float((got - want).abs().max())
        assert err <= 5e-5 * max(1.0, float(want.abs().max())), (soft, err)


def test_reference_symbol_SplatForwardGpu(cuda):
    """The reference's own C entry point (splat_gpu_impl.cuh:11-22: stream first, void) called through ctypes exactly as
    splat_gpu.c:29-31 calls it: bitwise equal to gg_splat_forward_f32 (binned boxes: no atomics) and - when the
    compiled reference kernel travelled to this box - equal to ITS SplatForwardGpu within float-atomic reordering."""
    import ctypes
    from gangealing_amd import _lib
    lib = _lib.load()
    lib.SplatForwardGpu.restype = None
    lib.SplatForwardGpu.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5
    n, c, h, w, p = 2, 3, 96, 80, 3000
    g = torch.Generator().manual_seed(11)
    coords = (torch.rand(n, p, 2, generator=g) * torch.tensor([w + 8.0, h + 8.0]) - 4.0).to(cuda)
    values = torch.randn(n, p, c, generator=g).to(cuda)
    sigma = torch.tensor([1.3, 2.5]).to(cuda)
    stream = torch.cuda.current_stream().cuda_stream

    def run(fn, stream_first):
        alpha = torch.zeros(n, h, w, device=cuda)
        out = torch.zeros(n, c, h, w, device=cuda)
        ptrs = [coords.data_ptr(), values.data_ptr(), sigma.data_ptr(), alpha.data_ptr(), out.data_ptr()]
        if stream_first:
            fn(stream, *ptrs, p, c, h, w, n * p)
        else:
            assert fn(*ptrs, p, c, h, w, n * p, stream) == 0
        torch.cuda.synchronize()
        return alpha, out

    a1, o1 = run(lib.SplatForwardGpu, True)
    a0, o0 = run(lib.gg_splat_forward_f32, False)
    assert torch.equal(a1, a0) and torch.equal(o1, o0)
    assert float(a1.abs().max()) > 0
    from oracle import make_golden_splat as ref
    if ref.reference_available():
        rl = ctypes.CDLL(ref.REF_SO)
        rl.SplatForwardGpu.restype = None
        rl.SplatForwardGpu.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5
        ar, orr = run(rl.SplatForwardGpu, True)
        assert float((a1 - ar).abs().max()) <= 2e-5 * max(1.0, float(ar.abs().max()))
        assert float((o1 - orr).abs().max()) <= 2e-5 * max(1.0, float(orr.abs().max()))


@pytest.mark.parametrize('case', [
    dict(n=2, c=3, h=96, w=130, p=4000, sigma=(1.3, 0.4), tag='two sigmas, ragged tiles'),
    dict(n=1, c=6, h=64, w=64, p=1500, sigma=(2.0,), tag='two channel passes'),
    dict(n=1, c=2, h=160, w=96, p=600, sigma=(12.0,), tag='boxes beyond 33 pixels: the atomic path'),
    dict(n=2, c=1, h=128, w=128, p=800, sigma=(7.9, 9.5), tag='binned and large boxes in one call'),
    dict(n=1, c=3, h=64, w=64, p=12000, sigma=(1.0,), crowd=True, tag='> 8192 points in one tile: global sort'),
], ids=lambda c: c['tag'])
def test_splat2d_binned_gather_against_restatement(case, cuda):
    """Round 4 formulation (bins per 32x32 tile, lists sorted by point index, one gather block per tile, normalisation
    fused into the single write per pixel) against the numpy restatement of the reference semantics (float64 sums),
    on the cases that exercise its branches; and two calls are BITWISE identical whenever no box exceeds 33 pixels."""
    from gangealing_amd.splat2d_cuda import splat2d
    from oracle import np_ops
    n, c, h, w, p = case['n'], case['c'], case['h'], case['w'], case['p']
    g = torch.Generator().manual_seed(p + c)
    if case.get('crowd'):
        coords = torch.rand(n, p, 2, generator=g) * 20.0 + 5.0               # everything inside tile (0, 0)
    else:
        coords = torch.rand(n, p, 2, generator=g) * torch.tensor([w + 8.0, h + 8.0]) - 4.0
    values = torch.randn(n, p, c, generator=g)
    sigma = torch.tensor(case['sigma'], dtype=torch.float32)
    inp = torch.randn(n, c, h, w, generator=g) * 0.1
    for soft in (False, True):
        want = np_ops.splat2d(inp.numpy(), coords.numpy(), values.numpy(), sigma.numpy(), soft)
        got = splat2d(inp.to(cuda), coords.to(cuda), values.to(cuda), sigma.to(cuda), soft)
        # un-normalised pixels that no point reached are input / 1e-8: compare relative to each plane's largest value
        err = float(np.abs(got.cpu().numpy() - want).max() / max(1.0, float(np.abs(want).max())))
        assert err <= 2e-5, (soft, err)
        again = splat2d(inp.to(cuda), coords.to(cuda), values.to(cuda), sigma.to(cuda), soft)
        if max(case['sigma']) <= 8.0:
            assert torch.equal(got, again), 'binned gather must be bitwise reproducible'


def test_splat_forward_accumulates_like_the_reference_entry_point(cuda):
    """gg_splat_forward_f32 = SplatForwardGpu (splat_gpu_impl.cuh:11-22): ADDS alpha and alpha * value onto what the
    two buffers hold (the torch glue pre-fills them with zeros / a copy of the input)."""
    from gangealing_amd import _lib
    g = torch.Generator().manual_seed(3)
    n, p, c, h, w = 2, 700, 3, 70, 45
    coords = (torch.rand(n, p, 2, generator=g) * torch.tensor([w * 1.0, h * 1.0])).to(cuda)
    values = torch.randn(n, p, c, generator=g).to(cuda)
    sigma = torch.tensor([1.1, 2.3]).to(cuda)
    alpha0 = torch.rand(n, h, w, generator=g).to(cuda)
    out0 = torch.randn(n, c, h, w, generator=g).to(cuda)
    alpha, out = alpha0.clone(), out0.clone()
    _lib.call('gg_splat_forward_f32', coords, values, sigma, alpha, out, p, c, h, w, n * p)
    za, zo = torch.zeros_like(alpha0), torch.zeros_like(out0)
    _lib.call('gg_splat_forward_f32', coords, values, sigma, za, zo, p, c, h, w, n * p)
    assert float(za.max()) > 0
    assert float((alpha - (alpha0 + za)).abs().max()) <= 1e-5 * float(za.max())
    assert float((out - (out0 + zo)).abs().max()) <= 1e-5 * float(zo.abs().max())


def test_splat2d_errors(cuda):
    from gangealing_amd.splat2d_cuda import splat2d
    with pytest.raises(NotImplementedError):
        splat2d(torch.zeros(1, 1, 4, 4), torch.zeros(1, 2, 2), torch.zeros(1, 2, 1), torch.ones(1))
    out = splat2d(torch.zeros(1, 1, 4, 4, device=cuda, requires_grad=True), torch.ones(1, 2, 2, device=cuda),
                  torch.ones(1, 2, 1, device=cuda), torch.ones(1, device=cuda))
    with pytest.raises(NotImplementedError):
        out.sum().backward()
    # empty point set / points all out of bounds leave input / (0 + 1e-8) untouched-shaped output
    out = splat2d(torch.zeros(1, 2, 4, 4, device=cuda), torch.full((1, 3, 2), -5.0, device=cuda),
                  torch.ones(1, 3, 2, device=cuda), torch.ones(1, device=cuda))
    assert float(out.abs().max()) == 0.0


# ----------------------------------------------------------------------------- convolutions

CONV_CASES = [
    # n, cin, cout, h, k, stride, pad, groups, transposed
    (2, 8, 16, 9, 3, 1, 1, 1, False),
    (3, 64, 128, 17, 3, 1, 1, 1, False),
    (2, 16, 8, 17, 3, 2, 0, 1, False),          # STN downsample conv (after blur)
    (2, 16, 24, 15, 1, 2, 0, 1, False),         # STN skip conv
    (2, 3, 64, 16, 1, 1, 0, 1, False),
    (1, 512, 3, 8, 1, 1, 0, 1, False),          # ToRGB shape (narrow tile)
    (2, 12, 10, 4, 3, 2, 0, 1, True),           # generator up-conv
    (1, 24, 18, 8, 3, 1, 1, 3, False),          # grouped
    (1, 24, 16, 5, 3, 2, 0, 2, True),           # grouped transposed (per-sample form)
    (2, 8, 8, 6, 3, 1, 1, 1, True),             # transposed stride 1
    (2, 130, 140, 20, 3, 1, 1, 1, False),       # ragged tiles in both dims
]


@pytest.mark.parametrize('spec', CONV_CASES, ids=lambda s: 'x'.join(map(str, s)))
def test_conv_vs_torch_cpu(spec, cuda):
    import torch.nn.functional as F
    from gangealing_amd.op import conv2d_gradfix
    n, cin, cout, h, k, stride, pad, groups, transposed = spec
    gen = torch.Generator().manual_seed(hash(spec) & 0xFFFF)
    x = torch.randn(n, cin, h, h + 1, generator=gen)
    if transposed:
        w = torch.randn(cin, cout // groups, k, k, generator=gen) / (cin * k * k) ** 0.5
    else:
        w = torch.randn(cout, cin // groups, k, k, generator=gen) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=gen)
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xg, wg, bg = (t.to(cuda).requires_grad_(True) for t in (x, w, b))
    if transposed:
        ref = F.conv_transpose2d(xc, wc, bc, stride=stride, padding=pad, groups=groups)
        out = conv2d_gradfix.conv_transpose2d(xg, wg, bg, stride=stride, padding=pad, groups=groups)
    else:
        ref = F.conv2d(xc, wc, bc, stride=stride, padding=pad, groups=groups)
        out = conv2d_gradfix.conv2d(xg, wg, bg, stride=stride, padding=pad, groups=groups)
    assert out.shape == ref.shape
    close(out, ref.detach().numpy(), TOL)
    g = torch.randn(ref.shape, generator=gen)
    ref.backward(g)
    out.backward(g.to(cuda))
    close(xg.grad, xc.grad.numpy(), TOL)
    close(wg.grad, wc.grad.numpy(), 2e-4, 2e-4)
    close(bg.grad, bc.grad.numpy(), 2e-4, 2e-4)


@pytest.mark.parametrize('case', load_golden('modulated_conv'))
def test_modulated_conv_golden(case, cuda):
    """Shared-weight modulated conv == the reference's per-sample grouped ModulatedConv2d (+ Blur)."""
    from gangealing_amd.stylegan2.networks import ModulatedConv2d
    m = case['meta']
    mod = ModulatedConv2d(m['cin'], m['cout'], m['k'], 12, demodulate=m['demodulate'], upsample=m['upsample']).to(cuda)
    with torch.no_grad():
        mod.weight.copy_(T(case['weight'], cuda)[None])
        mod.modulation.weight.copy_(T(case['mod_weight'], cuda))
        mod.modulation.bias.copy_(T(case['mod_bias'], cuda))
    mod.requires_grad_(False)            # generator weights are frozen on this path (train.py:64-65)
    x, w = T(case['x'], cuda, True), T(case['w'], cuda, True)
    out = mod(x, w)
    close(out, case['out'], TOL)
    out.backward(T(case['g'], cuda))
    close(x.grad, case['gx'], TOL)
    close(w.grad, case['gw'], 2e-4, 2e-4)


@pytest.mark.parametrize('n,style_dim,cin,cout', [(16, 512, 512, 512), (3, 512, 128, 3), (5, 96, 70, 33)])
def test_style_demod_matches_torch_ops(n, style_dim, cin, cout, cuda):
    """EqualLinear modulation + demodulation in one launch == the reference's op sequence (networks.py:214-249)."""
    from gangealing_amd.op.conv_mfma import style_demod
    g = torch.Generator(device='cpu').manual_seed(5)
    wplus = torch.randn(n, 4, style_dim, generator=g).to(cuda)
    latent = wplus[:, 2]                                          # a strided W+ slot
    w = torch.randn(cin, style_dim, generator=g).to(cuda)
    b = torch.randn(cin, generator=g).to(cuda)
    wsq = torch.rand(cout, cin, generator=g).to(cuda)
    scale, lr_mul = style_dim ** -0.5, 1.0
    style, demod = style_demod(latent, w, b, scale, lr_mul, wsq, 1e-8)
    ref_style = torch.nn.functional.linear(latent.double(), w.double() * scale, b.double() * lr_mul)
    ref_demod = torch.rsqrt((ref_style * ref_style) @ wsq.double().t() + 1e-8)
    np.testing.assert_allclose(style.cpu().numpy(), ref_style.cpu().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(demod.cpu().numpy(), ref_demod.cpu().numpy(), rtol=2e-5, atol=1e-7)
    style_only, none = style_demod(latent, w, None, scale, lr_mul)
    assert none is None
    np.testing.assert_allclose(style_only.cpu().numpy(), (ref_style - b.double() * lr_mul).cpu().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('n,c,h,w,use_lin', [(3, 64, 20, 28, False), (2, 512, 8, 8, True), (1, 7, 5, 3, False)])
def test_lpips_tail_matches_torch_ops(n, c, h, w, use_lin, cuda):
    """Fused perceptual-loss tail (normalize -> diff^2 -> lin / channel sum -> spatial mean) and its gradient against
    the reference's op sequence (lpips.py:26-28, 190-199) evaluated in float64."""
    from gangealing_amd.losses import lpips_tail
    g = torch.Generator(device='cpu').manual_seed(3)
    feats = torch.relu(torch.randn(2 * n, c, h, w, generator=g) + 0.3)
    lin = torch.rand(1, c, 1, 1, generator=g) if use_lin else None
    gout = torch.randn(n, generator=g)
    x = feats.to(cuda).requires_grad_(True)
    out = lpips_tail(x, None if lin is None else lin.to(cuda))
    out.backward(gout.to(cuda))
    xr = feats.double().requires_grad_(True)
    u = xr / (torch.sqrt(torch.sum(xr ** 2, dim=1, keepdim=True)) + 1e-10)
    d = (u[:n] - u[n:]) ** 2
    if lin is not None:
        d = d * lin.double()
    ref = d.sum(dim=1).mean(dim=(1, 2))
    ref.backward(gout.double())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-4, atol=2e-7 * float(xr.grad.abs().max()) * 50)


def test_torgb_dgrad_add_matches_autograd(cuda):
    """g += data gradient of the modulated 1x1 ToRGB convolution (gg_torgb_dgrad_add_f32) against autograd through
    the reference's formulation (networks.py:243-281 with demodulate=False)."""
    from gangealing_amd import _lib
    g0 = torch.Generator(device='cpu').manual_seed(4)
    n, c, h, w = 3, 40, 12, 20
    x = torch.randn(n, c, h, w, generator=g0, dtype=torch.float64, requires_grad=True)
    weight = torch.randn(3, c, generator=g0, dtype=torch.float64)
    style = torch.randn(n, c, generator=g0, dtype=torch.float64)
    grad_rgb = torch.randn(n, 3, h, w, generator=g0, dtype=torch.float64)
    scale = c ** -0.5
    rgb = torch.einsum('nkc,nchw->nkhw', weight[None] * scale * style[:, None, :], x)
    (ref,) = torch.autograd.grad(rgb, x, grad_rgb)
    running = torch.randn(n, c, h, w, generator=g0)
    g = running.clone().to(cuda)
    _lib.call('gg_torgb_dgrad_add_f32', g, grad_rgb.float().to(cuda), weight.float().to(cuda), style.float().to(cuda),
              scale, n, c, h * w)
    np.testing.assert_allclose(g.cpu().numpy(), (running.double() + ref).numpy(), rtol=1e-5, atol=1e-5)


# ----------------------------------------------------------------------------- round 2: launch-tail fusions

def test_upfirdn2d_add_matches_separate_ops(cuda):
    """ToRGB's `rgb + Upsample(skip)` in one kernel: values and both gradients equal the two-op form."""
    from gangealing_amd.op.upfirdn2d import upfirdn2d, upfirdn2d_add
    g = torch.Generator().manual_seed(1)
    k = torch.tensor(np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0 * 4, dtype=torch.float32, device=cuda)
    for (n, c, h) in [(2, 3, 8), (3, 3, 64), (1, 5, 17)]:
        skip = torch.randn(n, c, h, h, generator=g).to(cuda).requires_grad_(True)
        rgb = torch.randn(n, c, 2 * h, 2 * h, generator=g).to(cuda).requires_grad_(True)
        gout = torch.randn(n, c, 2 * h, 2 * h, generator=g).to(cuda)
        a = upfirdn2d_add(skip, k, rgb, up=2, down=1, pad=(2, 1))
        ga = torch.autograd.grad(a, (skip, rgb), gout)
        b = upfirdn2d(skip, k, up=2, down=1, pad=(2, 1)) + rgb
        gb = torch.autograd.grad(b, (skip, rgb), gout)
        torch.testing.assert_close(a, b, atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(ga[0], gb[0], atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(ga[1], gb[1], atol=0, rtol=0)


def test_add_scale_matches_torch(cuda):
    from gangealing_amd.op.conv_mfma import add_scale
    g = torch.Generator().manual_seed(2)
    for shape in [(2, 7, 5, 3), (4, 64, 32, 32), (1, 1, 1, 1)]:
        a = torch.randn(*shape, generator=g).to(cuda).requires_grad_(True)
        b = torch.randn(*shape, generator=g).to(cuda).requires_grad_(True)
        y = add_scale(a, b, 2 ** -0.5)
        ga, gb = torch.autograd.grad(y, (a, b), torch.ones_like(y))
        torch.testing.assert_close(y, (a + b) * 2 ** -0.5, atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(ga, torch.full_like(a, 2 ** -0.5))
        torch.testing.assert_close(gb, torch.full_like(b, 2 ** -0.5))


def test_style_bank_matches_per_layer_modulation(cuda):
    """All modulation / demodulation vectors of a generator pass from two launches == the per-layer computation
    (networks.py:214-216,244-249), for the whole W+ and for the slots behind the learned ones."""
    from gangealing_amd.stylegan2 import Generator
    torch.manual_seed(3)
    g = Generator(64, 512, 8).to(cuda).eval().requires_grad_(False)
    latent = torch.randn(5, g.n_latent, 512, device=cuda)
    for first in (0, 3):
        pre = g._styles(latent, first)
        layers = [(m, s) for m, s in g._layer_slots() if s >= first]
        assert len(pre) == len(layers) > 0
        for m, slot in layers:
            style, demod = pre[m]
            ref_style = torch.nn.functional.linear(latent[:, slot].double(), m.modulation.weight.double() * m.modulation.scale,
                                                   m.modulation.bias.double() * m.modulation.lr_mul)
            np.testing.assert_allclose(style.cpu().numpy(), ref_style.cpu().numpy(), rtol=2e-5, atol=2e-5)
            if m.demodulate:
                w = m.weight[0].double() * m.scale
                ref_demod = torch.rsqrt(ref_style.pow(2) @ w.pow(2).sum(dim=(2, 3)).t() + 1e-8)
                np.testing.assert_allclose(demod.cpu().numpy(), ref_demod.cpu().numpy(), rtol=3e-5, atol=1e-7)
            else:
                assert demod is None
    # a latent that needs gradients everywhere leaves nothing to the bank
    assert g._styles(latent.clone().requires_grad_(True), g.n_latent) == {}


def test_generator_fusions_do_not_change_the_image(cuda):
    """Style bank, noise bank (explicit noise given here), ToRGB bias / skip fusion on vs off: same image, same
    latent gradient."""
    from gangealing_amd.op import conv_mfma
    from gangealing_amd.stylegan2 import Generator
    torch.manual_seed(4)
    g = Generator(64, 512, 8).to(cuda).eval().requires_grad_(False)
    with torch.no_grad():
        for p in g.parameters():
            if p.dim() == 4 and p.shape[1] == 3:       # ToRGB biases (1, 3, 1, 1) are zero-initialised
                p.normal_(0, 0.3)
    noise = [torch.randn(3, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), device=cuda) for i in range(g.num_layers)]
    w = torch.randn(3, g.n_latent, 512, device=cuda)

    def run(disabled):
        old = conv_mfma.DISABLED
        conv_mfma.DISABLED = frozenset(disabled)
        try:
            lat = w.clone().requires_grad_(True)
            img, _ = g([lat], input_is_latent=True, noise=noise, grad_latents=4)
            (gl,) = torch.autograd.grad(img.square().mean(), lat)
            return img.detach(), gl
        finally:
            conv_mfma.DISABLED = old
    img_on, g_on = run(())
    img_off, g_off = run(('style_bank', 'torgb_bias', 'noise_bank'))
    # (batch 3 at 64^2: every layer splits K with fp32 atomics, so two runs of the SAME configuration already differ in
    # the last bits, and 14 layers amplify that: typical difference 1e-6, occasionally above 2e-5)
    assert float((img_on - img_off).norm() / img_off.norm()) < 2e-5
    torch.testing.assert_close(img_on, img_off, atol=2e-4, rtol=1e-4)
    # the two runs differ by fp32 rounding (different kernels carry the bias / the adds), which the split-K atomics'
    # run-to-run ordering also produces; a leaky-ReLU input that lands on the other side of zero turns that into a
    # visible gradient entry, so the gradient is compared in L2 with a bounded worst entry
    scale = float(g_off.abs().max())
    assert float((g_on - g_off).norm() / g_off.norm()) < 1e-3
    assert float((g_on - g_off).abs().max()) < 1e-6 + 5e-3 * scale
    assert float(g_on[:, 4:].abs().max()) == 0.0 and float(g_on[:, :4].abs().max()) > 0


def test_lpips_tap_accumulates_into_the_downstream_gradient(cuda):
    """The tap node (features passed on + distance) gives the same value and the same feature gradient as the
    separate tail with autograd's addition."""
    from gangealing_amd.losses import lpips_tail, lpips_tap
    g = torch.Generator().manual_seed(5)
    f = torch.relu(torch.randn(6, 64, 16, 16, generator=g) + 0.2).to(cuda)
    lin = torch.rand(64, generator=g).to(cuda)
    wnext = torch.randn(6, 64, 16, 16, generator=g).to(cuda)
    gv = torch.randn(3, generator=g).to(cuda)
    for use_lin in (None, lin):
        fa = f.clone().requires_grad_(True)
        passed, val = lpips_tap(fa * 1.0, use_lin)
        (ga,) = torch.autograd.grad([(passed * wnext).sum(), val], fa, [torch.ones((), device=cuda), gv])
        fb = f.clone().requires_grad_(True)
        fb1 = fb * 1.0
        (gb,) = torch.autograd.grad([(fb1 * wnext).sum(), lpips_tail(fb1, use_lin)], fb, [torch.ones((), device=cuda), gv])
        torch.testing.assert_close(val, lpips_tail(f, use_lin), atol=1e-6, rtol=1e-5)
        torch.testing.assert_close(ga, gb, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('shape', [(2, 128, 16, 16, 3), (3, 512, 4, 4, 3), (2, 96, 8, 12, 4), (1, 33, 6, 6, 1),
                                   (2, 64, 10, 10, 2)])
def test_conv1x1_few_outputs_streaming_kernel(shape, cuda):
    """ToRGB-shaped 1x1 convolutions (<= 4 outputs) run the streaming channel-reduction kernel: against F.conv2d on the
    CPU, with per-sample input / output scales and a bias, and against the MFMA path's answer for 5 outputs."""
    import torch.nn.functional as F
    from gangealing_amd.op import conv_mfma
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    s_in, s_out = torch.rand(n, cin, generator=g) + 0.5, torch.rand(n, cout, generator=g) + 0.5
    ref = F.conv2d(x.double() * s_in.double()[:, :, None, None], wt.double()) * s_out.double()[:, :, None, None] \
        + b.double()[None, :, None, None]
    wm = conv_mfma.PackedWeight(wt.to(cuda), 1, cout, cin, 1, 0, 0)
    y = conv_mfma.conv_forward(x.to(cuda), wm, n, 1, cin, cout, 1, 1, 0, 0, in_scale=s_in.to(cuda), out_scale=s_out.to(cuda),
                               bias=b.to(cuda))
    assert float((y.double().cpu() - ref).abs().max()) < 2e-6 * max(1.0, float(ref.abs().max()))
    y0 = conv_mfma.conv2d(x.to(cuda), wt.to(cuda), None)
    assert float((y0.double().cpu() - F.conv2d(x.double(), wt.double())).abs().max()) < 2e-6 * max(1.0, float(ref.abs().max()))


def test_unsplit_launches_are_bitwise_reproducible(cuda):
    """Launches that do not split K (no float atomics) must give bit-identical results run after run: the wave-level
    LDS staging of the epilogues and the double-buffered, software-pipelined tap loop have no cross-wave hazards."""
    from gangealing_amd.op import conv_mfma
    old = conv_mfma.PRECISION
    conv_mfma.set_precision('bf16x3')
    try:
        g = torch.Generator().manual_seed(5)
        cases = [  # (n, cin, cout, h, k, stride, pad, mode, scaled)
            (32, 128, 128, 64, 3, 1, 1, 0, True),      # 512 tiles of 256 pixels: pipelined loop
            (16, 128, 128, 64, 3, 1, 1, 0, False),     # 512 tiles of 128 pixels
            (16, 64, 64, 64, 3, 1, 1, 0, False),       # narrow (64-channel) tile
            (16, 256, 256, 64, 3, 2, 0, 1, True),      # transposed, 128-q tiles (1088 blocks)
            (12, 128, 128, 32, 3, 2, 0, 1, False),     # transposed, 64-q tiles (216 blocks: unsplit)
        ]
        for (n, cin, cout, h, k, stride, pad, mode, scaled) in cases:
            x = torch.randn(n, cin, h, h, generator=g).to(cuda)
            w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(cuda)
            wm = conv_mfma.PackedWeight(w, 1, cout, cin, k, 0, 0)
            s_in = (torch.rand(n, cin, generator=g) + 0.5).to(cuda) if scaled else None
            s_out = (torch.rand(n, cout, generator=g) + 0.5).to(cuda) if scaled else None
            first = conv_mfma.conv_forward(x, wm, n, 1, cin, cout, k, stride, pad, mode, in_scale=s_in, out_scale=s_out)
            for _ in range(30):
                again = conv_mfma.conv_forward(x, wm, n, 1, cin, cout, k, stride, pad, mode, in_scale=s_in, out_scale=s_out)
                assert torch.equal(first, again), (n, cin, cout, h, mode)
    finally:
        conv_mfma.set_precision(old)


@pytest.mark.parametrize('shape', [(2, 3, 8, 8), (1, 2, 6, 10), (3, 5, 16, 32), (2, 4, 2, 2)])
def test_maxpool2x2_matches_aten(shape, cuda):
    """VGG16's 2x2 max pooling (torchvision features -> ATen max_pool2d): values, the tie rule (first maximum in
    row-major order keeps the gradient) and NaN propagation against F.max_pool2d on the CPU, forward and backward."""
    import torch.nn.functional as F
    from gangealing_amd.losses import max_pool2x2
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    x = (x * 2).round() / 2                       # half-integer values: many exact ties inside windows
    x[0, 0, 0, 1] = float('nan')
    gy = torch.randn(shape[0], shape[1], shape[2] // 2, shape[3] // 2, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 2, 2)
    ref.backward(gy)
    xc = x.to(cuda).requires_grad_(True)
    out = max_pool2x2(xc)
    out.backward(gy.to(cuda))
    assert torch.equal(torch.nan_to_num(out.detach().cpu(), nan=123.0), torch.nan_to_num(ref.detach(), nan=123.0))
    assert torch.equal(xc.grad.cpu(), xr.grad)


def test_repack_many_equals_single_packs(cuda):
    """gg_conv_pack_weights_many (one launch re-packing every trainable weight after the optimizer step; LDS-staged
    transposes) writes bit for bit what the per-weight entry point writes, for both layouts, flips, 1x1 / 3x3, ragged
    channel counts and groups."""
    from gangealing_amd import _lib
    dt = np.dtype([('dst', '<u8'), ('src', '<u8'), ('total', '<i8'), ('limb_stride', '<i8'), ('cout_g', '<i4'),
                   ('cin_g', '<i4'), ('kh', '<i4'), ('kw', '<i4'), ('transpose_io', '<i4'), ('flip', '<i4'),
                   ('limbs', '<i4'), ('scale', '<f4')])
    g = torch.Generator().manual_seed(21)
    specs = [  # groups, cout_g, cin_g, k, transpose_io, flip, limbs, scale
        (1, 512, 512, 3, 0, 0, 2, 0.5), (1, 512, 512, 3, 1, 1, 2, 0.5), (1, 128, 64, 3, 1, 0, 2, 1.0),
        (1, 128, 64, 1, 0, 0, 2, 1.0), (1, 64, 128, 1, 1, 0, 2, 0.25), (2, 40, 96, 3, 1, 1, 3, 1.0),
        (2, 40, 96, 3, 0, 1, 1, 1.0), (1, 576, 512, 3, 1, 1, 2, 1.0), (1, 33, 70, 3, 1, 0, 2, 1.0),
        (1, 64, 3, 3, 0, 0, 2, 1.0),
    ]
    keep, rows, singles = [], [], []
    for (groups, cout_g, cin_g, k, tio, flip, limbs, scale) in specs:
        shape = (groups * cin_g, cout_g, k, k) if tio else (groups * cout_g, cin_g, k, k)
        w = torch.randn(*shape, generator=g).to(cuda)
        n = groups * cout_g * cin_g * k * k
        many = torch.zeros((limbs, n), dtype=torch.int16, device=cuda)
        one = torch.zeros((limbs, n), dtype=torch.int16, device=cuda)
        _lib.call('gg_conv_pack_weight_split', one, w, groups, cout_g, cin_g, k, k, tio, flip, scale, limbs)
        rows.append((many.data_ptr(), w.data_ptr(), n, n, cout_g, cin_g, k, k, tio, flip, limbs, scale))
        keep.append((w, many))
        singles.append(one)
    jobs = torch.from_numpy(np.array(rows, dtype=dt).view(np.uint8).reshape(-1).copy()).to(cuda)
    _lib.call('gg_conv_pack_weights_many', jobs, len(rows))
    for spec, (_, many), one in zip(specs, keep, singles):
        assert torch.equal(many, one), spec


def test_torch_library_ops_run_the_hip_kernels(cuda):
    """torch.ops.gangealing.{upfirdn2d, fused_leaky_relu, splat2d, mipmap_warp} == the module-level operators,
    including autograd through the registered formulas."""
    import gangealing_amd.op.library  # noqa: F401
    from gangealing_amd.op import upfirdn2d, fused_leaky_relu
    from gangealing_amd.splat2d_cuda import splat2d
    g = torch.Generator().manual_seed(9)
    k = torch.tensor(np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0, dtype=torch.float32, device=cuda)
    x = torch.randn(2, 3, 9, 9, generator=g).to(cuda).requires_grad_(True)
    gy = torch.randn(2, 3, 18, 18, generator=g).to(cuda)
    a = torch.ops.gangealing.upfirdn2d(x, k * 4, 2, 1, 2, 1)
    b = upfirdn2d(x, k * 4, up=2, down=1, pad=(2, 1))
    torch.testing.assert_close(a, b, atol=0, rtol=0)
    torch.testing.assert_close(torch.autograd.grad(a, x, gy)[0], torch.autograd.grad(b, x, gy)[0], atol=1e-6, rtol=1e-6)
    bias = torch.randn(3, generator=g).to(cuda).requires_grad_(True)
    a = torch.ops.gangealing.fused_leaky_relu(x, bias, 0.2, 2 ** 0.5)
    b = fused_leaky_relu(x, bias, 0.2, 2 ** 0.5)
    torch.testing.assert_close(a, b, atol=0, rtol=0)
    ga, gb = torch.autograd.grad(a, (x, bias), torch.ones_like(a)), torch.autograd.grad(b, (x, bias), torch.ones_like(b))
    torch.testing.assert_close(ga[0], gb[0])
    torch.testing.assert_close(ga[1], gb[1])
    coords = (torch.rand(2, 11, 2, generator=g) * 9).to(cuda)
    vals = torch.randn(2, 11, 3, generator=g).to(cuda)
    sigma = torch.tensor([1.0, 1.7], device=cuda)
    img = torch.zeros(2, 3, 9, 9, device=cuda)
    torch.testing.assert_close(torch.ops.gangealing.splat2d(img, coords, vals, sigma, False),
                               splat2d(img, coords, vals, sigma, False), atol=1e-6, rtol=1e-5)
    # convolutions: dispatcher op == module function, gradients through the Autograd-key registration
    from gangealing_amd.op import conv_mfma
    xc = torch.randn(2, 32, 16, 16, generator=g).to(cuda).requires_grad_(True)
    wc = (torch.randn(64, 32, 3, 3, generator=g) * 0.1).to(cuda).requires_grad_(True)
    bc = torch.randn(64, generator=g).to(cuda).requires_grad_(True)
    a = torch.ops.gangealing.conv2d(xc, wc, bc, 2, 1, 1)
    b = conv_mfma.conv2d(xc, wc, bc, stride=2, padding=1)
    torch.testing.assert_close(a, b, atol=0, rtol=0)              # (split-K partial sums are added in a fixed order: bitwise)
    gy = torch.randn_like(a)
    for u, v in zip(torch.autograd.grad(a, (xc, wc, bc), gy), torch.autograd.grad(b, (xc, wc, bc), gy)):
        torch.testing.assert_close(u, v, atol=1e-5, rtol=1e-5)
    wt = (torch.randn(32, 48, 3, 3, generator=g) * 0.1).to(cuda).requires_grad_(True)
    a = torch.ops.gangealing.conv_transpose2d(xc, wt, None, 2, 0, 0, 1)
    b = conv_mfma.conv_transpose2d(xc, wt, None, stride=2, padding=0)
    assert a.shape == (2, 48, 33, 33)
    torch.testing.assert_close(a, b, atol=0, rtol=0)              # same kernels, fixed summation order: bitwise
    gy = torch.randn_like(a)
    for u, v in zip(torch.autograd.grad(a, (xc, wt), gy), torch.autograd.grad(b, (xc, wt), gy)):
        torch.testing.assert_close(u, v, atol=0, rtol=0)
    # mipmap_warp: differentiable through the dispatcher op as through the module (round-2 ADVICE: the op silently
    # detached its outputs); splat2d's backward raises, as the reference's
    from gangealing_amd.spatial_transformers.antialiased_sampling import MipmapWarp
    from gangealing_amd.spatial_transformers.flow_ops import affine_grid
    img = torch.randn(2, 3, 32, 32, generator=g).to(cuda).requires_grad_(True)
    theta = torch.tensor([[[0.9, 0.1, 0.05], [-0.1, 0.8, -0.02]], [[1.6, 0.0, 0.0], [0.0, 1.7, 0.1]]], device=cuda)
    grid = affine_grid(theta, (2, 3, 24, 24)).detach().requires_grad_(True)
    out_op, lev_op = torch.ops.gangealing.mipmap_warp(img, grid, 2.5, 0.0, 'reflection', True)   # max_level = max_num_levels - 1
    out_mod = MipmapWarp(3.5)(img, grid, padding_mode='reflection')
    torch.testing.assert_close(out_op, out_mod, atol=0, rtol=0)
    assert out_op.requires_grad                     # (round 2: silently detached; the level map's gradient is ignored)
    gy = torch.randn_like(out_op)
    for u, v in zip(torch.autograd.grad(out_op, (img, grid), gy), torch.autograd.grad(out_mod, (img, grid), gy)):
        torch.testing.assert_close(u, v, atol=1e-6, rtol=1e-5)     # (the image gradient is a scatter with float atomics)
    canvas = torch.zeros(2, 3, 9, 9, device=cuda, requires_grad=True)
    with pytest.raises(NotImplementedError):
        torch.ops.gangealing.splat2d(canvas, coords, vals, sigma, False).sum().backward()


def test_splat_points_overlay_against_reference_kernel(cuda):
    """helpers.splat_points (mixed-reality / propagation overlay: two splats + alpha composite) against the same
    lines evaluated on the compiled reference kernel (when it travelled to this box) and against the restatement."""
    from gangealing_amd.splat2d_cuda.overlay import splat_points
    from oracle import make_golden_splat as ref
    from oracle import np_ops
    g = torch.Generator().manual_seed(12)
    n, p, h, w = 2, 300, 48, 64
    images = (torch.rand(n, 3, h, w, generator=g) * 2 - 1).to(cuda)
    points = (torch.rand(n, 2, p // 2, 2, generator=g) * torch.tensor([w + 6.0, h + 6.0]) - 3.0).to(cuda)
    colors = (torch.rand(n, p, 3, generator=g) * 2 - 1).to(cuda)
    alpha = torch.rand(n, p, 1, generator=g).to(cuda)
    out = splat_points(images, points, 1.4, 0.8, colors=colors, alpha_channel=alpha)
    assert out.shape == images.shape
    pts = points.reshape(n, p, 2).cpu().numpy()
    sig = np.full((n,), 1.4, np.float32)
    obj = np_ops.splat2d(np.zeros((n, 3, h, w), np.float32), pts, colors.cpu().numpy(), sig, False)
    mask = np_ops.splat2d(np.zeros((n, 1, h, w), np.float32), pts, alpha.cpu().numpy(), sig, True) * 0.8
    np.testing.assert_allclose(out.cpu().numpy(), mask * obj + (1 - mask) * images.cpu().numpy(), atol=2e-5, rtol=1e-4)
    if ref.reference_available():
        want = ref.reference_splat_points(images, points, 1.4, 0.8, colors, alpha)
        assert float((out - want).abs().max()) <= 5e-5
    with pytest.raises(NotImplementedError):
        splat_points(images, points, 1.4, 0.8)                       # colour-scale lookup is not part of the package
