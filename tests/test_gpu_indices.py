"""north_star: "bit-exact warp grid indices".  The sampling kernels' integer by-products - floor(ix), floor(iy) after
unnormalise + padding-mode transform, floor / ceil of the clamped mip level, and the stack depth D the reference
would build - are read back through gg_mipmap_warp_indices_f32 (the same device functions the forward / backward
kernels call) and compared with np.array_equal on EVERY output pixel against integers produced by ATen's formulas and
the reference's own level code (oracle/make_golden.py::gen_warp_indices): grids that land exactly on integers,
half-integers and the reflection seams (each +-1 ulp), far out-of-range coordinates, and neighbour distances in the
ulp-neighbourhood of 1, 2 and 4 (mip level exactly integral and the floats beside it).  No point is masked out."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

CASES = load_golden('warp_indices')


@pytest.mark.parametrize('case', CASES, ids=lambda c: f"{c['meta']['name']}-{c['meta']['padding_mode']}")
def test_indices_and_levels_bit_exact(case, cuda):
    from gangealing_amd.spatial_transformers.antialiased_sampling import warp_indices, MipmapWarp
    m = case['meta']
    grid = torch.from_numpy(case['grid']).to(cuda)
    ix, iy, lo, hi = warp_indices(grid, m['size'], m['size'], m['max_num_levels'], m['min_level'], m['padding_mode'])
    for name, got in (('ix_nw', ix), ('iy_nw', iy), ('level_floor', lo), ('level_ceil', hi)):
        got = got.cpu().numpy()
        ref = case[name]
        bad = np.argwhere(got != ref)
        assert np.array_equal(got, ref), (name, len(bad), bad[:5].tolist(),
                                          [(int(got[tuple(b)]), int(ref[tuple(b)])) for b in bad[:5]])
    assert int(hi.max()) + 1 == m['num_levels']                  # D of antialiased_sampling.py:52
    # the fractional level the forward kernel reports agrees with the reference's float level
    x = torch.zeros((grid.shape[0], 1, m['size'], m['size']), device=cuda)
    warp = MipmapWarp(max_num_levels=m['max_num_levels'])
    warp(x, grid, min_level=m['min_level'], padding_mode=m['padding_mode'])
    np.testing.assert_allclose((warp.levels_map * (m['max_num_levels'] - 1.0)).cpu().numpy(), case['levels'], atol=2e-6)


def test_plain_warp_indices_have_level_zero(cuda):
    from gangealing_amd.spatial_transformers.antialiased_sampling import warp_indices
    case = CASES[0]
    grid = torch.from_numpy(case['grid']).to(cuda)
    ix, iy, lo, hi = warp_indices(grid, 32, 32, padding_mode=case['meta']['padding_mode'], antialias=False)
    assert np.array_equal(ix.cpu().numpy(), case['ix_nw']) and np.array_equal(iy.cpu().numpy(), case['iy_nw'])
    assert int(lo.abs().max()) == 0 and int(hi.abs().max()) == 0
