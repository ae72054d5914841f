"""Why the perceptual loss's input gradient sits 1e-3 (not 1e-6) from the reference, demonstrated: the VGG16 trunk is
piecewise linear, and a unit whose pre-activation lies within rounding distance of zero (or a pooling window whose two
largest entries are that close) takes the other branch in another implementation; everything downstream inherits the
change.  tests/golden/lpips_masks.npz holds the REFERENCE's own branch decisions for the run behind lpips.npz (one bit
per ReLU unit, two per max-pool window; oracle/make_golden_configs.py lpips_masks).  Here the HIP path

  1. counts how many of its own decisions differ from the reference's (a handful among 6.6 M), and
  2. re-runs with the reference's decisions PINNED - every flipped unit's activation is moved to the reference's side
     of the kink by a denormal-sized edit of the saved output - and must then reproduce the reference's gradient to
     rounding accuracy: <= 1e-5 relative L2 with the exact-product kernels (1e-4 on two bf16 limbs), orders of
     magnitude below the un-pinned distance (measured on MI355X: fp32 kernels 1.8e-3 with ONE flipped ReLU among
     6.6 M units -> 3.5e-6 pinned; bf16x3 3.9e-3 with 11 ReLU + 4 pooling flips -> 5.7e-5 pinned).  So the kernels' backward arithmetic is exact to rounding; the 1e-3 is entirely branch flips.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, record_parity, PARITY

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['fp32', 'bf16x3', 'fp16x3'])
def mode(request):
    from gangealing_amd.op import conv_mfma
    old = conv_mfma.PRECISION
    conv_mfma.set_precision(request.param)
    yield request.param
    conv_mfma.set_precision(old)


def unpack_masks(case, device):
    shapes = case['meta']['shapes']
    relu, pool = {}, {}
    for idx in case['meta']['relu_layers']:
        shp = shapes[f'relu{idx}']
        bits = np.unpackbits(case[f'relu{idx}'])[:int(np.prod(shp))].reshape(shp).astype(bool)
        relu[idx] = torch.from_numpy(bits).to(device)
    for idx in case['meta']['pool_layers']:
        shp = shapes[f'pool{idx}']
        b = case[f'pool{idx}']
        codes = np.stack([b & 3, (b >> 2) & 3, (b >> 4) & 3, (b >> 6) & 3], 1).reshape(-1)[:int(np.prod(shp))]
        pool[idx] = torch.from_numpy(codes.reshape(shp).astype(np.int64)).to(device)
    return relu, pool


def winner_codes(x):
    """ATen's max_pool2d winner of every 2x2 window (first of equal maxima in row-major order): dy * 2 + dx."""
    c = [x[..., dy::2, dx::2] for dy in (0, 1) for dx in (0, 1)]
    best, code = c[0], torch.zeros_like(c[0], dtype=torch.int64)
    for k in (1, 2, 3):
        upd = c[k] > best
        code = torch.where(upd, torch.full_like(code, k), code)
        best = torch.where(upd, c[k], best)
    return code


class Pinner:
    """observe() callback for the VGG16 trunk: counts disagreements with the reference's branch decisions and, when
    `pin`, edits the activation in place (through .data: the tensor autograd saved IS this one) so that the backward
    takes the reference's branches.  Edits are denormal-sized for ReLU flips (0 <-> 1e-30) and one ulp above the window
    maximum for pooling flips."""
    POOL_AFTER = {2: 4, 7: 9, 14: 16, 21: 23}       # conv index -> the max-pool that consumes its output

    def __init__(self, relu, pool, pin):
        self.relu, self.pool, self.pin = relu, pool, pin
        self.relu_flips, self.pool_flips, self.units = 0, 0, 0

    def __call__(self, idx, y):
        ref = self.relu[idx]
        assert tuple(ref.shape) == tuple(y.shape), (idx, ref.shape, y.shape)
        data = y.data
        ours = data > 0
        self.units += ref.numel()
        self.relu_flips += int((ours != ref).sum())
        if self.pin:
            data[ref & ~ours] = 1e-30              # reference active, ours exactly 0: tiny positive -> gradient passes
            data[~ref & ours] = 0.0                # reference inactive, ours barely positive
        pidx = self.POOL_AFTER.get(idx)
        if pidx is not None:
            want = self.pool[pidx]
            have = winner_codes(data)
            diff = have != want
            self.pool_flips += int(diff.sum())
            if self.pin and bool(diff.any()):
                views = [data[..., dy::2, dx::2] for dy in (0, 1) for dx in (0, 1)]
                top = torch.stack(views, 0).max(0).values
                bump = torch.nextafter(top, torch.full_like(top, float('inf')))
                for k in range(4):
                    sel = diff & (want == k)
                    views[k][sel] = bump[sel]      # strided views of `data`: written through
                assert bool((winner_codes(data) == want).all())


@pytest.mark.parametrize('lp', [False, True], ids=['baseline', 'lin'])
def test_lpips_gradient_is_exact_once_branch_decisions_are_pinned(lp, mode, cuda):
    from oracle import config_cases as cc
    from gangealing_amd.losses import LPIPS
    case = next(c for c in load_golden('lpips') if c['meta']['lpips'] == lp)
    (masks,) = load_golden('lpips_masks')
    relu, pool = unpack_masks(masks, cuda)
    net = LPIPS(net='vgg', lpips=lp, pnet_rand=True, pretrained=False)
    torch.nn.Module.load_state_dict(net, cc.det_lpips_state_dict(net), strict=False)
    net = net.to(cuda).eval()
    test = f"lpips_masks[{'lin' if lp else 'baseline'}]"
    ref32 = np.asarray(case['gin0'], dtype=np.float64)
    ref64 = np.asarray(case['gin0_64'], dtype=np.float64)
    out = {}
    for pin in (False, True):
        pinner = Pinner(relu, pool, pin)
        net.observe = pinner
        in0 = torch.from_numpy(case['in0']).to(cuda).requires_grad_(True)
        in1 = torch.from_numpy(case['in1']).to(cuda)
        val = net(in0, in1)
        err = record_parity(test, mode, 'val_pinned' if pin else 'val', val.detach().cpu().numpy(), case['val'])
        assert err <= 1e-4 * max(1.0, float(np.abs(case['val']).max()))         # pinning does not move the forward
        val.backward(torch.from_numpy(case['g']).to(cuda))
        g = in0.grad.double().cpu().numpy()
        out[pin] = dict(rel_l2_vs_reference_fp32=float(np.linalg.norm(g - ref32) / np.linalg.norm(ref32)),
                        rel_l2_vs_reference_fp64=float(np.linalg.norm(g - ref64) / np.linalg.norm(ref64)),
                        relu_flips=pinner.relu_flips, pool_flips=pinner.pool_flips, units=pinner.units)
    net.observe = None
    PARITY.setdefault(test, {}).setdefault(mode, {})['gin0'] = dict(free=out[False], pinned=out[True])
    # the decisions themselves: only a handful of the 6.6 M units sit close enough to a kink to flip
    assert out[False]['relu_flips'] + out[False]['pool_flips'] <= (16 if mode == 'fp32' else 128), out[False]
    # with the reference's decisions the gradient is the reference's, to the rounding of 13 layers of arithmetic
    # measured: fp32 3.5e-6, bf16x3 5.7e-5, fp16x3 9.3e-6 (the backward of fp16x3 runs on bf16 limbs; un-pinned: 1.8e-3 /
    # 3.9e-3 / 9.3e-6)
    bound = {'fp32': 1e-5, 'bf16x3': 1e-4, 'fp16x3': 2e-5}[mode]
    assert out[True]['rel_l2_vs_reference_fp32'] <= bound, out
    assert out[True]['rel_l2_vs_reference_fp64'] <= 2 * bound, out
    if mode == 'fp16x3':
        # the benched arithmetic on this fixture: its single differing ReLU decision is inconsequential, so even the
        # UN-pinned gradient is the reference's (the library is bitwise reproducible: this holds on every run)
        assert out[False]['rel_l2_vs_reference_fp32'] <= bound, out
