"""GPU parity, model level: the HIP generator / STN / training loss against the golden vectors that
the REFERENCE modules produced on CPU with the same name-keyed weights (oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def close(t, ref, atol=1e-4, rtol=1e-4):
    np.testing.assert_allclose(t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else t, ref, atol=atol, rtol=rtol)


def load_det(module, rules=()):
    from oracle.det_weights import det_state_dict
    torch.nn.Module.load_state_dict(module, det_state_dict(module, [tuple(r) for r in rules]), strict=False)
    return module


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_generator_golden(cuda, grad_tol=(2e-3, 1e-3)):
    from gangealing_amd.stylegan2 import Generator
    (c,) = load_golden('generator16')
    g = load_det(Generator(16, 512, 8)).to(cuda).eval().requires_grad_(False)
    noise = [T(c[f'noise{i}'], cuda) for i in range(c['meta']['num_layers'])]
    img, latent = g([T(c['z'], cuda)], return_latents=True, noise=noise)
    close(latent[:, 0], c['w'], 1e-5)
    close(img, c['img'], 1e-4)
    w = T(c['w'], cuda).requires_grad_(True)
    img2, _ = g([w.unsqueeze(1).repeat(1, g.n_latent, 1)], input_is_latent=True, noise=noise)
    close(img2, c['img_from_w'], 1e-4)
    img2.backward(T(c['gimg'], cuda))
    close(w.grad, c['gw'], *grad_tol)


@pytest.mark.parametrize('case', load_golden('stn'), ids=lambda c: '+'.join(c['meta']['transforms']))
def test_stn_golden(case, cuda):
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    from gangealing_amd.losses import flow_losses
    m = case['meta']
    stn = get_stn(m['transforms'], flow_size=m['flow_size'], supersize=m['supersize'], channel_multiplier=0.5,
                  num_heads=1)
    stn = load_det(stn, m['scale_rules']).to(cuda)
    x = T(case['x'], cuda)
    kw = dict(return_flow=True, padding_mode=m['padding_mode'])
    if m['supersize'] > m['flow_size']:
        kw['input_img_for_sampling'] = x
    out, fm = stn(x, **kw)
    close(out, case['out'], 1e-4)
    close(fm, case['flow_or_matrix'], 1e-4)
    loss = (out ** 2).mean()
    if 'flow' in m['transforms']:
        reg = flow_losses(fm)
        loss = loss + 10.0 * reg[0] + reg[1]
    close(loss, case['loss'], 1e-5)
    loss.backward()
    grads = dict((n, p.grad) for n, p in stn.named_parameters())
    worst = 0.0
    for name, ref_norm in m['grad_norms'].items():
        if ref_norm is None:
            continue
        got = float(grads[name].double().norm())
        worst = max(worst, abs(got - ref_norm) / max(ref_norm, 1e-9))
    # A similarity warp makes the 4 neighbour distances of MipmapWarp's level selection EXACTLY tied in
    # real arithmetic (antialiased_sampling.py:62-97); torch.max then routes the level gradient to whichever
    # neighbour wins by last-ulp rounding noise.  The sampling kernel reproduces that arg-max bit for bit on
    # identical grids (scripts/mipmap_golden_report.py: 2e-7), but here the grid comes out of our own
    # affine_grid / conv kernels (1-ulp different from ATen's), so ties break differently: a different,
    # equally valid sub-gradient.  Hence the looser bound on the similarity stage when it is followed by
    # level-gradient-carrying stages; the reference's own CUDA path differs from its CPU path the same way.
    assert worst < (2e-2 if len(m['transforms']) > 1 else 5e-3), worst
    for key in [k for k in case if k.startswith('grad_')]:
        name = key[len('grad_'):]
        match = [n for n in grads if n.replace('.', '_') == name]
        assert len(match) == 1
        if len(m['transforms']) > 1 and match[0].startswith('stns.0.'):
            continue            # see the note above
        close(grads[match[0]], case[key], 2e-4, 2e-3)


def test_train_step_golden(cuda, grad_rel=5e-3):
    """gangealing_loss + TV + identity (loss.py:64-75, train.py:117-124) with explicit z / noise."""
    from oracle.det_weights import det_array
    from gangealing_amd.stylegan2 import Generator
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    from gangealing_amd.latent_learner import DirectionInterpolator
    from gangealing_amd.losses import flow_losses
    (c,) = load_golden('train_step')
    m = c['meta']
    gen = load_det(Generator(m['gen_size'], 512, 8)).to(cuda).eval().requires_grad_(False)
    stn = get_stn(['similarity', 'flow'], flow_size=m['flow_size'], supersize=m['gen_size'], channel_multiplier=0.5,
                  num_heads=1)
    stn = load_det(stn, m['scale_rules']).to(cuda)
    ll = DirectionInterpolator(None, m['ndirs'], m['inject'], gen.n_latent).to(cuda)
    D = lambda name, shape, s=1.0: T(det_array(name, shape, s), cuda)
    with torch.no_grad():
        ll.directions.copy_(D('ll.directions', (m['ndirs'], 512)))
        ll.lat_mean.copy_(D('ll.lat_mean', (1, 512)))
        ll.coefficients.copy_(D('ll.coefficients', (1, m['ndirs']), 0.3))
    res = lambda i: 2 ** ((i + 5) // 2)
    n1 = [D(f'ts.n1.{i}', (2, 1, res(i), res(i))) for i in range(gen.num_layers)]
    n2 = [D(f'ts.n2.{i}', (2, 1, res(i), res(i))) for i in range(gen.num_layers)]
    with torch.no_grad():
        unaligned, w = gen([T(c['z'], cuda)], noise=n1, return_latents=True)
    target, _ = gen(ll([w[:, 0, :]], psi=m['psi']), input_is_latent=True, noise=n2)
    pred, delta = stn(unaligned, return_flow=True, padding_mode=m['padding_mode'])
    close(unaligned, c['unaligned'], 1e-4)
    close(target, c['target'], 1e-4)
    close(pred, c['pred'], 1e-4)
    close(delta, c['delta'], 1e-4)
    ploss = ((pred - target) ** 2).mean(dim=(1, 2, 3)).mean()
    reg = flow_losses(delta)
    total = ploss + m['tv_weight'] * reg[0] + m['flow_identity_weight'] * reg[1]
    close(total, c['total'], 1e-4)
    total.backward()
    close(ll.coefficients.grad, c['g_coefficients'], 1e-4 * grad_rel / 5e-3, 2e-3 * grad_rel / 5e-3)
    worst = 0.0
    grads = dict((n, p.grad) for n, p in stn.named_parameters())
    for name, ref_norm in m['grad_norms'].items():
        if name == 'll.coefficients':
            continue
        worst = max(worst, abs(float(grads[name].double().norm()) - ref_norm) / max(ref_norm, 1e-9))
    assert worst < grad_rel, worst


def test_trainer_step_runs_and_updates(cuda):
    from gangealing_amd.train_step import GangealingTrainer
    tr = GangealingTrainer(cuda, gen_size=64, flow_size=64, batch=2, inject=3, ndirs=2, perturb_heads=0.02)
    p0 = tr.stn_arena.param.clone()
    e0 = tr.ema_arena.param.clone()
    parts = tr.step(psi=0.5)
    g = tr.stn_arena.grad
    assert torch.isfinite(parts['p']) and torch.isfinite(g).all() and float(g.abs().max()) > 0
    # Adam's first step moves every parameter with a non-zero gradient by ~lr
    moved = (tr.stn_arena.param - p0).abs()
    nz = g.abs() > 1e-5          # |g| >> eps, so the first Adam step has magnitude lr
    assert float(moved[nz].max()) <= 1.001e-3 and float(moved[nz].min()) > 0.9e-3 * 0.5
    # EMA: ema = decay * ema + (1 - decay) * p
    d = tr.ema_decay
    torch.testing.assert_close(tr.ema_arena.param, e0 * d + tr.stn_arena.param * (1 - d), atol=1e-6, rtol=1e-5)
    # module parameters are views of the arena
    first = next(tr.stn.parameters())
    assert first.data_ptr() == tr.stn_arena.param.data_ptr()
    parts = tr.step(psi=0.4)
    assert torch.isfinite(parts['p'])


def test_adam_ema_kernel_vs_torch(cuda):
    from gangealing_amd import _lib
    torch.manual_seed(0)
    n = 100003
    p = torch.randn(n, device=cuda)
    g = torch.randn(n, device=cuda)
    ema = p.clone()
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    m = torch.zeros(n, device=cuda)
    v = torch.zeros(n, device=cuda)
    ema_ref = ema.clone()
    for step in range(1, 4):
        ref.grad = g.clone() * step
        opt.step()
        ema_ref.mul_(0.99).add_(ref.detach(), alpha=0.01)
        _lib.call('gg_adam_ema_f32', p, m, v, ema, (g * step).contiguous(), n, 1e-3, 0.9, 0.999, 1e-8, step, 0.99, 1.0)
    torch.testing.assert_close(p, ref.detach(), atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(ema, ema_ref, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize('cfg', [
    dict(name='c5-clustering', gen_size=64, flow_size=64, batch=2, num_heads=4, flips=True, inject=6, ndirs=5,
         sample_from_full_res=True),
    dict(name='c4-fullres', gen_size=128, flow_size=64, batch=2, inject=6, ndirs=8, padding_mode='border',
         sample_from_full_res=True),
    dict(name='c1-similarity', gen_size=64, flow_size=64, batch=4, transform=('similarity',), tv_weight=0.0,
         flow_identity_weight=0.0),
], ids=lambda c: c['name'])
def test_trainer_other_configs(cfg, cuda):
    """BASELINE.json configs 0/3/4 in miniature: K=4 clustering with flips (cartesian heads, min over 2K),
    full-resolution sampling through the mip pyramid, similarity-only STN."""
    from gangealing_amd.train_step import GangealingTrainer
    cfg = dict(cfg)
    cfg.pop('name')
    tr = GangealingTrainer(cuda, perturb_heads=0.02, **cfg)
    p0 = tr.stn_arena.param.clone()
    for psi in (0.6, 0.5):
        parts = tr.step(psi=psi)
    torch.cuda.synchronize()
    assert torch.isfinite(parts['p']).all()
    assert torch.isfinite(tr.stn_arena.grad).all() and torch.isfinite(tr.stn_arena.param).all()
    assert float((tr.stn_arena.param - p0).abs().max()) > 0
    assert float(tr.ll_arena.grad.abs().max()) > 0          # the latent learner receives a gradient through G


def test_cluster_stn_shapes(cuda):
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    stn = get_stn(['similarity', 'flow'], flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=3).to(cuda)
    x = torch.randn(2, 3, 64, 64, device=cuda)
    out, flow = stn(x, return_flow=True, padding_mode='reflection')
    assert out.shape == (6, 3, 64, 64) and flow.shape == (6, 64, 64, 2)
    # zero-initialised heads -> identity warp for every head (warping_heads.py:28-30,163-165)
    torch.testing.assert_close(out, x.repeat_interleave(3, dim=0), atol=1e-5, rtol=1e-5)
    out_u = stn(x, unfold=True, padding_mode='border')
    assert out_u.shape == (2, 3, 3, 64, 64)


@pytest.mark.parametrize('case', load_golden('cluster_classifier'), ids=lambda c: f"heads{c['meta']['num_heads']}")
def test_cluster_classifier_golden(case, cuda):
    """ResnetClassifier on the HIP kernels against the reference module: logits, loss, gradients, and the
    index-valued helpers (assign / run / run_flip / run_flip_cartesian) bit-exact."""
    from gangealing_amd.cluster_classifier import ResnetClassifier, accuracy
    m = case['meta']
    net = ResnetClassifier(m['size'], channel_multiplier=m['channel_multiplier'], num_heads=m['num_heads'],
                           supersize=m['supersize'])
    net = load_det(net, m['scale_rules']).to(cuda)
    x = T(case['x'], cuda)
    logits = net(x)
    close(logits, case['logits'], 1e-4)
    loss = torch.nn.functional.cross_entropy(logits, T(case['labels'], cuda))
    close(loss, case['loss'], 1e-5)
    loss.backward()
    grads = dict((n, p.grad) for n, p in net.named_parameters())
    close(grads['to_logits.weight'], case['grad_to_logits_weight'], 1e-5, 2e-3)
    close(grads['to_logits.bias'], case['grad_to_logits_bias'], 1e-5, 2e-3)
    for name, ref_norm in m['grad_norms'].items():
        got = float(grads[name].double().norm())
        assert abs(got - ref_norm) <= 5e-3 * max(ref_norm, 1e-9), (name, got, ref_norm)
    scores = T(case['scores'], cuda)
    assert float(accuracy(logits, scores)) == float(case['acc1'])
    assert float(accuracy(logits, scores, k=2)) == float(case['acc2'])
    with torch.no_grad():
        assert np.array_equal(net.assign(x).cpu().numpy(), case['assign'])
        assert np.array_equal(net.assign(x, ignore_flips=True).cpu().numpy(), case['assign_noflip'])
        flipped, _, classes, flip_ixs = net.run_flip(x)
        assert np.array_equal(classes.cpu().numpy(), case['run_flip_classes'])
        assert np.array_equal(flip_ixs.cpu().numpy(), case['run_flip_ixs'])
        assert np.array_equal(flipped.cpu().numpy(), case['run_flip_images'])
        kept, kept_logits = net.run(x, 1)
        assert np.array_equal(kept.cpu().numpy(), case['run_kept'])
        close(kept_logits, case['run_kept_logits'], 1e-4)
        cart, policy = net.run_flip_cartesian(x[:2])
        assert np.array_equal(cart.cpu().numpy(), case['cart_images'])
        assert np.array_equal(policy.cpu().numpy(), case['cart_policy'])


def test_cluster_classifier_training_step(cuda):
    """One iteration of train_cluster_classifier.py:78-101 on a miniature clustering configuration: labels come from
    the frozen cluster STN, the classifier's parameters move, nothing else does."""
    from gangealing_amd.cluster_classifier import ResnetClassifier, cluster_classifier_step
    from gangealing_amd.train_step import GangealingTrainer
    tr = GangealingTrainer(cuda, gen_size=64, flow_size=32, batch=4, transform=('similarity',), inject=3, ndirs=2,
                           num_heads=2, flips=True, perturb_heads=0.05, seed=3)
    heads = 2
    cls = ResnetClassifier(32, channel_multiplier=0.5, num_heads=heads * 2).to(cuda)
    opt = torch.optim.Adam(cls.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in cls.parameters()]
    stn_before = tr.stn_arena.param.clone()
    xent, metrics = cluster_classifier_step(cls, tr.generator, tr.t_ema, tr.ll, tr.loss_fn, tr.resize_fake2stn, 4,
                                            tr.dim_latent, heads, True, cuda, sample_from_full_res=False,
                                            padding_mode='border')
    opt.zero_grad()
    xent.backward()
    opt.step()
    assert torch.isfinite(xent) and 0.0 <= float(metrics['acc@1']) <= float(metrics['acc@2']) <= 1.0
    assert abs(float(metrics['assignments'].sum()) - 1.0) < 1e-6 and metrics['assignments'].numel() == 4
    assert any(float((p.detach() - b).abs().max()) > 0 for p, b in zip(cls.parameters(), before))
    assert torch.equal(tr.stn_arena.param, stn_before)


@pytest.mark.parametrize('case', load_golden('point_transfer'), ids=lambda c: '+'.join(c['meta']['transforms']))
def test_point_transfer_golden(case, cuda):
    """Key-point helpers (uncongeal / congeal / transfer points, forward_with_flip, match_flows)
    against the reference STN's own outputs (pixel coordinates)."""
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    m = case['meta']
    stn = get_stn(m['transforms'], flow_size=m['flow_size'], supersize=m['supersize'], channel_multiplier=0.5,
                  num_heads=1)
    stn = load_det(stn, m['scale_rules']).to(cuda).eval()
    imgA, imgB = T(case['imgA'], cuda), T(case['imgB'], cuda)
    pts, pts_n = T(case['points'], cuda), T(case['points_norm'], cuda)
    kw = dict(padding_mode='border')
    with torch.no_grad():
        unc = stn.uncongeal_points(imgB, pts_n, **kw)
        close(unc, case['uncongealed'], 2e-2, 1e-4)                      # pixels
        ref = torch.from_numpy(case['congealed']).to(cuda)
        if 'flow' in m['transforms']:
            # Index work (spatial_transformer.py:655-672: arg-min over the squared distances to the H x W nodes of the
            # sampling grid, evaluated in float32 as |x|^2 + |y|^2 - 2<x, y>): node indices must EQUAL the reference's -
            # or the two picks must be a tie of that float32 evaluation.  Proven per point on the float64 distance
            # field of the grid the HIP path produced (parity of the grid itself: 'fwf_flow' below, <= 1e-4): the
            # reference's node and ours are both within the float32 formula's rounding (4 ulp of |x|^2 + |y|^2, the
            # quantity it cancels) of the true minimum.  Key points on source-pixel centres sit exactly midway between
            # two nodes of the half-resolution grid wherever the flow is ~0: exact ties in real arithmetic.
            # (a composed STN: the similarity stage maps the points in closed form, the flow stage searches its grid -
            # spatial_transformer.py:232-262; the two stages are run here as ComposedSTN.congeal_points runs them)
            sim, flo = stn.stns
            out0, warp0, pts0 = sim.congeal_points(imgA, pts, normalize_input_points=True, unnormalize_output_points=True,
                                                   iters=1, output_resolution=stn.stn_in_size, base_warp=None,
                                                   input_img_for_sampling=imgA, return_full=True, **kw)
            _, fm, con = flo.congeal_points(out0, pts0, normalize_input_points=True, unnormalize_output_points=False,
                                            iters=1, output_resolution=None, base_warp=warp0, input_img_for_sampling=imgA,
                                            return_full=True, **kw)
            con = con.long()
            assert torch.equal(con, stn.congeal_points(imgA, pts, **kw).long())
            grid = (fm + flo.identity_flow).double()                       # (N, H, W, 2)
            h, w = grid.shape[1], grid.shape[2]
            p64 = flo.normalize(pts0, imgA.size(-1), imgA.size(-1)).double()
            d2 = (grid.reshape(grid.shape[0], h * w, 1, 2) - p64.reshape(p64.shape[0], 1, -1, 2)).pow(2).sum(-1)
            dmin = d2.min(dim=1).values                                    # (N, P)
            flat = lambda idx: idx[..., 1] * w + idx[..., 0]               # unravel_index orders (x, y)
            d_ours = d2.gather(1, flat(con).unsqueeze(1)).squeeze(1)
            d_ref = d2.gather(1, flat(ref.long()).unsqueeze(1)).squeeze(1)
            tol = 4 * 2.0 ** -23 * (p64.pow(2).sum(-1) + grid.pow(2).sum(-1).reshape(grid.shape[0], -1).max(1).values[:, None])
            exact = (con == ref.long()).all(-1)
            assert bool(((d_ours - dmin) <= tol).all()), 'our pick is not a minimiser of the distance field'
            assert bool((exact | ((d_ref - dmin) <= tol)).all()), \
                ('a reference pick that is not a float32 tie of ours', (d_ref - dmin)[~exact].tolist(), tol[~exact].tolist())
            record = float(exact.float().mean())
            from conftest import PARITY
            PARITY.setdefault('point_transfer[' + '+'.join(m['transforms']) + ']', {})['congeal_points'] = dict(
                points=int(exact.numel()), equal_to_reference=int(exact.sum()), proven_float32_ties=int((~exact).sum()),
                exact_fraction=record)
        else:
            con = stn.congeal_points(imgA, pts, **kw).float()
            close(con, case['congealed'], 1e-3, 1e-4)                     # normalised coordinates
        tra = stn.transfer_points(imgA, imgB, pts, **kw)
        if 'flow' in m['transforms']:
            # transfer = congeal (node pick) then a bilinear lookup of imgB's grid at that node: where the pick equals the
            # reference's the transferred point must too (to the lookup's float32 accuracy); a tie moves it by one node
            # of imgB's grid, bounded by that grid's local stretch
            t_ref = T(case['transferred'], cuda)
            err = (tra - t_ref).abs().amax(-1)
            assert float(err[exact].max() if bool(exact.any()) else 0.0) <= 2e-2, err[exact].max()
            assert float(err.max()) <= 1.5
        else:
            assert float((tra - T(case['transferred'], cuda)).abs().max()) <= 2e-2
        if 'flow' in m['transforms']:
            out, warp, flow, inputs, flips = stn.forward_with_flip(imgA, return_flow=True, return_warp=True,
                                                                    return_inputs=True, return_flip_indices=True, **kw)
            assert np.array_equal(flips.cpu().numpy(), case['fwf_flips'])
            close(out, case['fwf_out'], 1e-4)
            close(warp, case['fwf_warp'], 1e-4)
            close(flow, case['fwf_flow'], 1e-4)
            _, _, pa2, pb2, pick = stn.match_flows(imgA, imgB, pts, pts.flip(1), **kw)
            assert np.array_equal(pick.cpu().numpy(), case['mf_pick'])
            close(pa2, case['mf_pointsA'], 1e-4)
            close(pb2, case['mf_pointsB'], 1e-4)


def test_propagate_object_splat(cuda):
    """propagate_object = uncongeal_points + in-bounds selection + splat2d (the only caller of the splat kernel in the
    reference, spatial_transformer.py:297-366).  Checked against the numpy splat restatement on the same moved points
    (parity of the splat itself is unpinned: the reference op has no CPU path)."""
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    from oracle import np_ops
    stn = get_stn(['similarity', 'flow'], flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=1)
    stn = load_det(stn, (('warp_head.linear', 0.05), ('flow_out.2', 0.05), ('mask_out', 0.5))).to(cuda).eval()
    g = torch.Generator(device='cpu').manual_seed(2)
    n, p = 2, 40
    target = torch.randn(n, 3, 64, 64, generator=g).to(cuda)
    pts = (torch.rand(n, p, 2, generator=g) * 2.4 - 1.2).to(cuda)             # some land outside the image
    vals = torch.rand(n, p, 3, generator=g).to(cuda)
    mvals = torch.ones(n, p, 1, device=cuda)
    sigma = torch.tensor([1.0, 1.5], device=cuda)
    with torch.no_grad():
        obj, mask = stn.propagate_object(pts, vals, mvals, target, sigma, mem_efficient=True, padding_mode='border')
        moved = stn.uncongeal_points(target, pts, normalize_input_points=False, unnormalize_output_points=True,
                                     padding_mode='border')
    assert obj.shape == (n, 3, 64, 64) and mask.shape == (n, 1, 64, 64)
    for i in range(n):
        r = moved[i].round()
        keep = ((r[:, 0] >= 0) & (r[:, 1] >= 0) & (r[:, 0] < 64) & (r[:, 1] < 64)).cpu().numpy()
        assert 0 < keep.sum() < p
        blank = np.zeros((1, 3, 64, 64), np.float32)
        ref_obj = np_ops.splat2d(blank, moved[i:i + 1].cpu().numpy()[:, keep], vals[i:i + 1].cpu().numpy()[:, keep],
                                 sigma[i:i + 1].cpu().numpy(), False)
        ref_mask = np_ops.splat2d(blank[:, :1], moved[i:i + 1].cpu().numpy()[:, keep],
                                  mvals[i:i + 1].cpu().numpy()[:, keep], sigma[i:i + 1].cpu().numpy(), True)
        close(obj[i:i + 1], ref_obj, 1e-4, 1e-4)
        close(mask[i:i + 1], ref_mask, 1e-4, 1e-4)


def test_trainer_shortcuts_are_consistent_across_optimizer_steps(cuda):
    """At every one of three consecutive optimizer steps the loss and the gradient arena computed with all
    caller-side shortcuts on (one-launch weight re-pack validated by version counters, wgrad accumulating into the
    arena, fused epilogues / tails, masked data gradient) equal those recomputed on the SAME parameters with the
    shortcuts off.  A stale weight pack after the raw-pointer Adam update would show up at step 2."""
    from gangealing_amd.op import conv_mfma
    from gangealing_amd.train_step import GangealingTrainer
    tr = GangealingTrainer(cuda, gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3,
                           ndirs=2, perturb_heads=0.02, seed=5, stn_lr=1e-4, ll_lr=1e-4)
    off = frozenset(('slots', 'pack_registry', 'style_demod', 'fuse_act', 'lpips_tail', 'mask_dgrad', 'wgrad_rows',
                     'torgb_fuse'))

    def loss_and_grad(disabled, seed):
        old = conv_mfma.DISABLED
        conv_mfma.DISABLED = disabled
        try:
            torch.manual_seed(seed)
            tr.stn_arena.zero_grad()
            tr.ll_arena.zero_grad()
            total, _ = tr.loss(0.5)
            total.backward()
            return float(total.detach()), tr.stn_arena.grad.clone(), tr.ll_arena.grad.clone()
        finally:
            conv_mfma.DISABLED = old

    for step in range(3):
        l_off, g_off, gl_off = loss_and_grad(off, 100 + step)
        l_on, g_on, gl_on = loss_and_grad(frozenset(), 100 + step)
        # (a stale pack or a lost accumulation is an O(1) relative error; two evaluations of the SAME configuration
        # differ by split-K atomics ordering, which leaky-ReLU sign flips and the perceptual distance's eps = 1e-10
        # normalisation occasionally blow up to 1e-3 of the largest gradient entry - hence L2 + a loose worst entry)
        assert abs(l_on - l_off) <= 2e-4 * abs(l_off), (step, l_on, l_off)
        assert float((g_on - g_off).norm()) <= 5e-3 * float(g_off.norm()), step
        assert float((g_on - g_off).abs().max()) <= 5e-2 * float(g_off.abs().max()), step
        assert float((gl_on - gl_off).abs().max()) <= 5e-2 * float(gl_off.abs().max()) + 1e-9, step
        torch.manual_seed(100 + step)
        tr.step(psi=0.5)                                   # optimizer + EMA + re-pack, shortcuts on
    assert torch.isfinite(tr.stn_arena.param).all()


def test_pipelined_optimizer_update_matches_immediate_update(cuda):
    """pipeline_update defers the STN's all-reduce + Adam + EMA + re-pack until the STN is next used (behind the next
    iteration's generator passes).  One step + flush must leave the same parameters as the immediate update, the
    second step must see the updated parameters, and nothing may be pending after flush()."""
    from gangealing_amd.train_step import GangealingTrainer
    kw = dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2,
              perturb_heads=0.02, seed=9)

    def run(pipelined):
        tr = GangealingTrainer(cuda, pipeline_update=pipelined, **kw)
        p0 = tr.stn_arena.param.clone()
        torch.manual_seed(77)
        tr.step(psi=0.5)
        if pipelined:
            assert tr._pending is not None and torch.equal(tr.stn_arena.param, p0)      # not applied yet
        tr.flush()
        assert tr._pending is None
        p1 = tr.stn_arena.param.clone()
        e1 = tr.ema_arena.param.clone()
        torch.manual_seed(78)
        parts = tr.step(psi=0.5)                  # the STN forward of this step applies nothing twice
        tr.flush()
        return p0, p1, e1, tr.stn_arena.param.clone(), float(parts['p'])

    a0, a1, ae, a2, la = run(False)
    b0, b1, be, b2, lb = run(True)
    assert torch.equal(a0, b0)

    # nothing in the step adds floating-point numbers in a run-dependent order (no atomics), and the deferred update
    # shows every forward the same parameters as the immediate one: bitwise equality
    assert float((a1 - a0).abs().max()) > 0
    assert torch.equal(a1, b1) and torch.equal(ae, be) and la == lb
    assert torch.equal(a2, b2)
    moved = float(((a2 - a1).abs() > 1e-6).float().mean())
    assert moved > 0.9, moved                                      # the second update was applied


def test_ema_network_forward_follows_the_optimizer(cuda):
    """The fused Adam/EMA kernel rewrites the arenas through raw pointers.  The EMA STN (the network that is actually
    evaluated: visualisation, cluster-classifier labels) is frozen, so its conv weight packs and scaled EqualLinear
    weights are cached per parameter version - a forward after a step must see the NEW weights.  Checked against a
    freshly built module loaded with the EMA state (no caches)."""
    from gangealing_amd.train_step import GangealingTrainer
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    kw = dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2,
              perturb_heads=0.05, seed=21)
    tr = GangealingTrainer(cuda, stn_lr=2e-3, **kw)
    tr.ema_decay = 0.0                                     # EMA = current parameters: a large, visible change per step
    x = torch.randn(2, 3, 64, 64, device=cuda) * 0.5
    with torch.no_grad():
        out0, flow0 = tr.t_ema(x, return_flow=True, padding_mode='reflection')
    for _ in range(2):
        tr.step(psi=0.5)
        tr.flush()
        with torch.no_grad():
            out1, flow1 = tr.t_ema(x, return_flow=True, padding_mode='reflection')
        fresh = get_stn(['similarity', 'flow'], flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=1).to(cuda)
        fresh.load_state_dict(tr.t_ema.state_dict())
        fresh.requires_grad_(False)
        with torch.no_grad():
            out_ref, flow_ref = fresh(x, return_flow=True, padding_mode='reflection')
        assert float((flow1 - flow0).abs().max()) > 1e-4           # the update is visible at all
        # (split-K partial sums are combined with float atomics: two forwards of the same weights agree to ~1e-6)
        torch.testing.assert_close(flow1, flow_ref, atol=3e-4, rtol=1e-3)
        torch.testing.assert_close(out1, out_ref, atol=3e-4, rtol=1e-3)
        out0, flow0 = out1, flow1


def test_grad_slots_only_inside_the_trainers_backward(cuda):
    """Weight gradients go straight into the arena only inside `with conv_mfma.grad_slots()` (the trainer's own
    backward).  torch.autograd.grad on the same graph - a diagnostic backward, a second loss - gets ordinary gradients
    and leaves the arena untouched."""
    from gangealing_amd.op import conv_mfma
    from gangealing_amd.train_step import GangealingTrainer
    tr = GangealingTrainer(cuda, gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3,
                           ndirs=2, perturb_heads=0.02, seed=4)
    weights = [(n, p) for n, p in tr.stn.named_parameters() if p.dim() == 4]
    assert len(weights) > 10 and all(p.data_ptr() in conv_mfma.GRAD_SLOTS for _, p in weights)
    assert not any(p.data_ptr() in conv_mfma.GRAD_SLOTS for p in tr.t_ema.parameters())     # frozen: never registered
    tr.stn_arena.zero_grad()
    torch.manual_seed(3)
    total, _ = tr.loss(0.5)
    grads = torch.autograd.grad(total, [p for _, p in weights], retain_graph=True)
    assert all(g is not None and float(g.abs().max()) > 0 for g in grads)
    assert float(tr.stn_arena.grad.abs().max()) == 0.0
    with conv_mfma.grad_slots():
        total.backward()
    for (name, p), g in zip(weights, grads):
        torch.testing.assert_close(p.grad, g, atol=2e-6 + 2e-3 * float(g.abs().max()), rtol=0, msg=name)


def test_trainer_schedule_and_checkpoint_round_trip(cuda):
    """train_iteration drives psi and the learning rates as train.py:92-97,129-132; state_dict()/load_state_dict()
    carry weights, Adam moments, step counts and scheduler state in the reference's checkpoint layout."""
    from gangealing_amd.train_step import GangealingTrainer
    kw = dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2,
              perturb_heads=0.02, seed=8, anneal_psi=2, period=1.5, tm=2, decay=0.5, stn_lr=1e-4, ll_lr=1e-4)
    # (1e-4: with randomly initialised G / VGG the reference's 1e-3 drives the random STN into its chaotic regime
    # within a few iterations - bench.py SYNTHETIC_LR - and the continuation check below turns into a coin toss)
    a = GangealingTrainer(cuda, **kw)
    psis, lrs = [], []
    for i in range(1, 6):
        lrs.append(a.t_sched.get_last_lr()[0])
        torch.manual_seed(50 + i)
        _, psi = a.train_iteration(i)
        psis.append(psi)
    assert psis[0] == pytest.approx(0.5) and psis[1] == pytest.approx(0.0, abs=1e-7) and psis[2:] == [0.0, 0.0, 0.0]
    assert lrs[:3] == [1e-4, 1e-4, 1e-4] and lrs[3] < 1e-4 and a.t_sched.get_last_lr()[0] != lrs[3]
    ckpt = a.state_dict()
    assert set(ckpt) == {'g_ema', 't', 't_ema', 't_optim', 't_sched', 'll', 'll_optim', 'll_sched'}
    assert len(ckpt['t_optim']['state']) == len(list(a.stn.parameters()))
    b = GangealingTrainer(cuda, **dict(kw, seed=99))                 # different initial weights
    assert b.load_state_dict(ckpt) is True
    for x, y in ((a.stn_arena, b.stn_arena), (a.ema_arena, b.ema_arena), (a.ll_arena, b.ll_arena)):
        assert torch.equal(x.param, y.param) and torch.equal(x.exp_avg, y.exp_avg) and \
            torch.equal(x.exp_avg_sq, y.exp_avg_sq) and x.step_count == y.step_count
    assert b.t_sched.get_last_lr() == a.t_sched.get_last_lr()
    # both continue identically (same data seed): same loss, same gradient, and an update of the same direction and
    # length (entries whose gradient is ~0 may take either sign under Adam, so no element-wise comparison)
    before = a.stn_arena.param.clone()
    torch.manual_seed(500)
    pa, _ = a.train_iteration(6)
    ga = a.stn_arena.grad.clone()
    torch.manual_seed(500)
    pb, _ = b.train_iteration(6)
    # (two evaluations of the same step agree to ~1e-5 in the loss in general, but the perceptual distance normalises
    # feature vectors with eps = 1e-10: a pixel whose features are all ~0 is a discontinuity - 0.5 % observed here)
    assert abs(float(pa['p']) - float(pb['p'])) <= 5e-2 * abs(float(pa['p']))
    da, db = (a.stn_arena.param - before).double(), (b.stn_arena.param - before).double()
    assert float((da * db).sum() / (da.norm() * db.norm())) > 0.5 and 0.5 < float(da.norm() / db.norm()) < 2.0


def test_graph_replay_trainer(cuda):
    """use_graph: the whole iteration captured in a hipGraph after the eager warm-up iterations.  Replays must train
    (finite losses, parameters move, EMA follows), the device-resident scalars must be live (a zero learning rate
    freezes the parameters, psi reaches the latent mix) and the host-side step counters / version counters must track
    what the replays did."""
    from gangealing_amd.train_step import GangealingTrainer
    kw = dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2,
              perturb_heads=0.02, seed=13)
    tr = GangealingTrainer(cuda, use_graph=True, graph_warmup=2, **kw)
    p0 = tr.stn_arena.param.clone()
    losses = []
    for i in range(6):
        parts = tr.step(psi=0.5)
        losses.append(float(parts['p']))
        assert (tr._graph is None) == (i < 2)
    assert all(np.isfinite(losses)) and tr.stn_arena.step_count == 6 and tr.ll_arena.step_count == 6
    p6 = tr.stn_arena.param.clone()
    assert float((p6 - p0).abs().max()) > 1e-3
    v = [p._version for p in tr.stn.parameters()][:3]
    # zero learning rate: the replay reads lr from device memory -> parameters stay, moments still update
    m_before = tr.stn_arena.exp_avg.clone()
    tr.step(psi=0.5, stn_lr=0.0, ll_lr=0.0)
    assert torch.equal(tr.stn_arena.param, p6) and not torch.equal(tr.stn_arena.exp_avg, m_before)
    assert [p._version for p in tr.stn.parameters()][:3] != v
    # psi is read from device memory too: psi = 1 makes the target the unaligned sample's own latent, psi = 0 the
    # mean latent - different losses from the same replayed graph and the same random stream
    torch.manual_seed(1)
    a = float(tr.step(psi=1.0, stn_lr=0.0, ll_lr=0.0)['p'])
    torch.manual_seed(1)
    b = float(tr.step(psi=0.0, stn_lr=0.0, ll_lr=0.0)['p'])
    torch.manual_seed(1)
    a2 = float(tr.step(psi=1.0, stn_lr=0.0, ll_lr=0.0)['p'])
    assert abs(a - a2) <= 2e-2 * abs(a) and abs(a - b) > 5e-2 * abs(a), (a, b, a2)
    # the EMA network (evaluated eagerly, outside the graph) sees the replayed updates
    x = torch.randn(2, 3, 64, 64, device=cuda) * 0.3
    with torch.no_grad():
        e1 = tr.t_ema(x, padding_mode='border')
    tr.step(psi=0.5)
    with torch.no_grad():
        e2 = tr.t_ema(x, padding_mode='border')
    assert float((e1 - e2).abs().max()) > 0


@pytest.mark.parametrize('case', load_golden('stn_inference'), ids=lambda c: '+'.join(c['meta']['transforms']))
def test_stn_inference_options_golden(case, cuda):
    """iters > 1 (iterated_forward, spatial_transformer.py:523-567), return_intermediates, output_resolution,
    return_out_of_bounds with and without image_bounds (warping_heads.py:280-310) against the reference modules."""
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    m = case['meta']
    stn = get_stn(m['transforms'], flow_size=64, supersize=128, channel_multiplier=0.5, num_heads=1)
    stn = load_det(stn, m['scale_rules']).to(cuda).eval()
    x = T(case['x'], cuda)
    with torch.no_grad():
        if len(m['transforms']) == 1:
            out3, grid3, m3, oob3 = stn(x, iters=3, return_warp=True, return_flow=True, return_out_of_bounds=True,
                                        output_resolution=96, padding_mode='border')
            assert np.array_equal(oob3.cpu().numpy(), case['oob3'])
            out1, oob1 = stn(x, return_out_of_bounds=True, padding_mode='reflection')
            close(out1, case['out1'], 1e-4)
            assert np.array_equal(oob1.cpu().numpy(), case['oob1'])
            bounds = T(case['bounds'], cuda)
            oob_b = torch.cat([stn(x[i:i + 1], return_out_of_bounds=True, padding_mode='border',
                                   image_bounds=bounds[i:i + 1])[1] for i in range(3)])
            assert np.array_equal(oob_b.cpu().numpy(), case['oob_b'])
            # all images at once (the reference's comparison only broadcasts for one image): same answers
            _, oob_all = stn(x, return_out_of_bounds=True, padding_mode='border', image_bounds=bounds)
            assert np.array_equal(oob_all.cpu().numpy(), case['oob_b'])
            outs, mats = stn(x, iters=3, return_intermediates=True, padding_mode='border')
        else:
            out3, grid3, m3 = stn(x, iters=3, return_warp=True, return_flow=True, output_resolution=96,
                                  padding_mode='border')
            outs, mats = stn(x, iters=2, return_intermediates=True, padding_mode='reflection')
        close(out3, case['out3'], 1e-4)
        close(grid3, case['grid3'], 2e-5)
        close(m3, case['m3'], 2e-5)
        close(torch.stack(outs), case['inter_out'], 1e-4)
        close(torch.stack(mats), case['inter_m'], 2e-5)
