"""CPU suite: host-side logic of the product (no HIP compute): drop-in signatures, state_dict layout,
latent learner math, launcher injection, flat arenas."""
import inspect
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO


def test_signatures_match_reference_contract():
    from gangealing_amd.op import upfirdn2d, fused_leaky_relu, FusedLeakyReLU, conv2d_gradfix
    from gangealing_amd.splat2d_cuda import splat2d, Splat2D  # noqa: F401
    from gangealing_amd.spatial_transformers.antialiased_sampling import MipmapWarp, Warp, BilinearDownsample
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(upfirdn2d) == ['input', 'kernel', 'up', 'down', 'pad']
    assert sig(fused_leaky_relu) == ['input', 'bias', 'negative_slope', 'scale']
    assert sig(FusedLeakyReLU.__init__)[1:] == ['channel', 'negative_slope', 'scale']
    assert sig(conv2d_gradfix.conv2d) == ['input', 'weight', 'bias', 'stride', 'padding', 'dilation', 'groups']
    assert sig(conv2d_gradfix.conv_transpose2d) == ['input', 'weight', 'bias', 'stride', 'padding', 'output_padding',
                                                    'groups', 'dilation']
    assert hasattr(conv2d_gradfix, 'no_weight_gradients') and conv2d_gradfix.enabled is True
    assert sig(MipmapWarp.forward)[1:] == ['inputs', 'grid', 'min_level', 'padding_mode']
    assert sig(Warp.forward)[1:] == ['inputs', 'grid', 'padding_mode']
    assert sig(BilinearDownsample.__init__)[1:] == ['stride', 'channels']
    m = FusedLeakyReLU(7)
    assert [n for n, _ in m.named_parameters()] == ['bias'] and m.bias.shape == (7,)
    b = BilinearDownsample(2, 3)
    assert set(dict(b.named_buffers())) == {'kernel_horz', 'kernel_vert'}
    np.testing.assert_allclose(b.kernel_horz[0, 0, 0].numpy(), [1 / 8, 3 / 8, 3 / 8, 1 / 8])


def test_no_weight_gradients_context():
    from gangealing_amd.op import conv2d_gradfix
    assert conv2d_gradfix.weight_gradients_disabled is False
    with conv2d_gradfix.no_weight_gradients():
        assert conv2d_gradfix.weight_gradients_disabled is True
    assert conv2d_gradfix.weight_gradients_disabled is False


def test_model_parameter_counts_and_keys():
    """Param counts measured on the reference by instantiation: G@256 30.03 M,
    STN sim+flow@128 43.05 M (K=4: 51.05 M), STN sim@64 22.31 M."""
    from gangealing_amd.stylegan2 import Generator
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    count = lambda m: sum(p.numel() for p in m.parameters())
    assert count(Generator(256, 512, 8)) == 30034338
    kw = dict(flow_size=128, supersize=256, channel_multiplier=0.5)
    assert count(get_stn(['similarity', 'flow'], num_heads=1, **kw)) == 43054278
    assert count(get_stn(['similarity', 'flow'], num_heads=4, **kw)) == 51052440
    stn = get_stn(['similarity'], flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=1)
    assert count(stn) == 22305924
    keys = set(stn.state_dict())
    assert {'convs.0.1.bias', 'final_linear.weight', 'warp_head.linear.weight', 'warp_head.one_hot'} <= keys
    # checkpoint loading drops the keys the reference drops (spatial_transformer.py:378-385,722-726)
    sd = stn.state_dict()
    sd['warp_head.rebias'] = torch.zeros(1)
    stn.load_state_dict(sd)


def test_direction_interpolator_matches_formula():
    from gangealing_amd.latent_learner import DirectionInterpolator
    torch.manual_seed(0)
    ll = DirectionInterpolator(None, 3, 5, 14, num_heads=2)
    with torch.no_grad():
        ll.coefficients.copy_(torch.randn(2, 3))
    w = torch.randn(4, 512)
    (out,) = ll([w], psi=0.3)
    assert out.shape == (8, 14, 512)
    target = ll.lat_mean + ll.coefficients @ ll.directions             # (K, 512)
    for n in range(4):
        for k in range(2):
            exp = target[k] + 0.3 * (w[n] - target[k])
            torch.testing.assert_close(out[2 * n + k, 0], exp)
            torch.testing.assert_close(out[2 * n + k, 4], exp)
            torch.testing.assert_close(out[2 * n + k, 5], w[n])


def test_flat_arena_views_and_zero_grad():
    from gangealing_amd.train_step import FlatArena
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    ref = [p.detach().clone() for p in net.parameters()]
    arena = FlatArena(net)
    assert arena.numel == sum(p.numel() for p in ref)
    for p, r in zip(net.parameters(), ref):
        torch.testing.assert_close(p.detach(), r)
    net(torch.randn(2, 5)).sum().backward()
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    torch.testing.assert_close(arena.grad, flat)
    assert float(arena.grad.abs().sum()) > 0
    arena.zero_grad()
    assert float(arena.grad.abs().sum()) == 0
    net(torch.randn(2, 5)).sum().backward()
    assert float(arena.grad.abs().sum()) > 0          # autograd still accumulates into the arena


def test_cosine_psi():
    from gangealing_amd.train_step import cosine_psi
    assert cosine_psi(0, 100) == 1.0 and abs(cosine_psi(50, 100) - 0.5) < 1e-12 and cosine_psi(100, 100) == 0.0


@pytest.mark.skipif(not os.path.isdir('/root/reference/models'), reason='needs the reference checkout')
def test_launcher_injects_ops_into_reference_imports():
    """In the authoring container: with the launcher's injection, the reference's own networks.py /
    warping_heads.py import OUR operator modules (and never trigger the CUDA JIT build)."""
    import subprocess
    code = (
        "import sys; sys.path.insert(0, %r); sys.dont_write_bytecode = True\n"
        "from gangealing_amd import launch\n"
        "launch.inject('/root/reference')\n"
        "import models.stylegan2.networks as n, models.spatial_transformers.warping_heads as wh\n"
        "import gangealing_amd.op as op, gangealing_amd.spatial_transformers.antialiased_sampling as aa\n"
        "assert n.upfirdn2d is op.upfirdn2d and n.FusedLeakyReLU is op.FusedLeakyReLU\n"
        "assert n.conv2d_gradfix is op.conv2d_gradfix and wh.MipmapWarp is aa.MipmapWarp\n"
        "print('ok')\n" % REPO)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-2000:]
