"""CPU suite: the numpy oracle (oracle/np_ops.py) against the golden vectors that
oracle/make_golden.py produced from the reference's own CPU bodies."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import np_ops as O


def close(a, b, atol, rtol=1e-5):
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


@pytest.mark.parametrize('case', load_golden('upfirdn2d'), ids=lambda c: c['meta']['tag'])
def test_upfirdn2d(case):
    m = case['meta']
    up, down, pad = (m['up'],) * 2, (m['down'],) * 2, tuple(m['pad'])
    out = O.upfirdn2d(case['x'], case['k'], up, down, pad)
    assert out.shape == case['out'].shape
    close(out, case['out'], 2e-6)
    gx = O.upfirdn2d_backward(case['g'], case['k'], up, down, pad, case['x'].shape)
    assert gx.shape == case['x'].shape
    close(gx, case['gx'], 2e-6)


@pytest.mark.parametrize('case', load_golden('fused_act'))
def test_fused_act(case):
    out = O.fused_leaky_relu(case['x'], case['b'])
    close(out, case['out'], 1e-6)
    gx, gb = O.fused_leaky_relu_backward(case['g'], out)
    # sign reference is the OUTPUT (fused_act.py:60): identical to autograd except where x+b == 0
    nz = (case['x'] + case['b'].reshape([1, -1] + [1] * (case['x'].ndim - 2))) != 0
    close(gx[nz], case['gx'][nz], 1e-6)
    if nz.all():
        close(gb, case['gb'], 1e-5)


@pytest.mark.parametrize('case', load_golden('grid_sample'), ids=lambda c: c['meta']['padding_mode'] + '-' + c['meta']['grid'])
def test_grid_sample(case):
    out = O.grid_sample(case['x'], case['grid'], case['meta']['padding_mode'])
    close(out, case['out'], 2e-6)


@pytest.mark.parametrize('case', load_golden('mipmap_warp'),
                         ids=lambda c: f"{c['x'].shape[-1]}-{c['meta']['padding_mode']}-{c['meta']['grid']}")
def test_mipmap_warp(case):
    m = case['meta']
    out, aux = O.mipmap_warp(case['x'], case['grid'], m['max_num_levels'], 0.0, m['padding_mode'], return_aux=True)
    close(out, case['out'], 5e-6)
    close(aux['levels'] / (m['max_num_levels'] - 1.0), case['levels_map'], 1e-6)
    # the integer floor/ceil levels from the exponent trick == floor/ceil of the reference's levels
    ref_levels = case['levels_map'] * np.float32(m['max_num_levels'] - 1.0)
    _, dmax = O.mip_levels(case['grid'], case['x'].shape[2], case['x'].shape[3], m['max_num_levels'])
    fl, ce = O.mip_level_ints(dmax, m['max_num_levels'])
    safe = np.abs(ref_levels - np.round(ref_levels)) > 1e-5        # reference value not within rounding of an integer
    assert (fl[safe] == np.floor(ref_levels[safe])).all()
    assert (ce[safe] == np.ceil(ref_levels[safe])).all()
    assert (fl == aux['level_floor']).all() and (ce == aux['level_ceil']).all()


def test_similarity_head():
    (c,) = load_golden('similarity_head')
    m = O.make_affine_matrix(c['params'])
    close(m, c['matrix'], 1e-6)
    comp = O.compose_affine(c['base'], m)
    close(comp, c['composed'], 1e-6)
    grid = O.affine_grid(comp, 16, 16)
    close(grid, c['grid'], 2e-6)


@pytest.mark.parametrize('case', load_golden('flow_head'))
def test_flow_head(case):
    close(O.identity_flow(case['identity'].shape[1]), case['identity'], 1e-7)
    flow, delta = O.flow_compose(case['low'], case['mask'], case['base'], 8)
    close(delta, case['delta'], 2e-6)
    close(flow, case['flow'], 2e-6)
    res = O.interpolate_bilinear(np.ascontiguousarray(case['flow'].transpose(0, 3, 1, 2)), 2.0).transpose(0, 2, 3, 1)
    close(res, case['resized2x'], 2e-6)


@pytest.mark.parametrize('case', load_golden('bilinear_downsample'))
def test_bilinear_downsample(case):
    close(O.bilinear_downsample(case['x'], case['meta']['stride']), case['out'], 2e-6)


@pytest.mark.parametrize('case', load_golden('flow_losses'))
def test_flow_losses(case):
    close(O.total_variation_loss(case['delta']), case['tv'], 1e-6)
    close(O.flow_identity_loss(case['delta']), case['identity'], 1e-6)


@pytest.mark.parametrize('case', load_golden('modulated_conv'))
def test_modulated_conv(case):
    m = case['meta']
    scale = 1.0 / np.sqrt(case['mod_weight'].shape[1])
    style = case['w'] @ (case['mod_weight'] * np.float32(scale)).T + case['mod_bias']
    close(style, case['style'], 1e-5)
    blur = (np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0 * 4).astype(np.float32)
    out = O.modulated_conv2d(case['x'], case['weight'], case['style'], m['demodulate'], m['upsample'], blur, (1, 1))
    close(out, case['out'], 2e-5, 1e-4)


@pytest.mark.parametrize('case', load_golden('splat2d'))
def test_splat2d_selfcheck(case):
    out = O.splat2d(case['input'], case['coords'], case['values'], case['sigma'], case['meta']['soft_normalize'])
    close(out, case['out'], 1e-6)
    assert np.isfinite(out).all()
