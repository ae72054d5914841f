"""SURVEY.md section 8 f4, inference glue: load_stn, determine_flips and the per-batch body of the mixed-reality loop
(gangealing_amd/applications.py) against the REFERENCE's own functions run on CPU (oracle/make_golden.py applications):
applications.determine_flips + STN.uncongeal_points + un-mirroring + crop offsets + the congealed frames, for a composed
STN on non-square frames without classifier ('unimodal') and a K = 2 clustering STN with its classifier ('predict_cluster'
one frame at a time).  Plus the 'fixed_cluster' mode (reference tensors are created with device='cuda' there: no CPU
golden) checked for consistency with the per-cluster calls, and the overlay through splat2d."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_models import load_det, T, close

pytestmark = pytest.mark.gpu


def build(case, cuda):
    from gangealing_amd import applications as app
    from gangealing_amd.cluster_classifier import ResnetClassifier
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    m = case['meta']
    args = types.SimpleNamespace(**m['args'])
    heads = m['num_heads']
    t = load_det(get_stn(args.transform, flow_size=args.flow_size, supersize=args.real_size,
                         channel_multiplier=args.stn_channel_multiplier, num_heads=heads), m['stn_rules'])
    ckpt = {'t_ema': t.state_dict()}
    if heads > 1:
        cls = load_det(ResnetClassifier(args.flow_size, channel_multiplier=0.5, num_heads=2 * heads,
                                        supersize=args.real_size), [('to_logits', 0.05)])
        ckpt['classifier'] = cls.state_dict()
    # through the loader, as an application would (checkpoint dict in the reference's layout, train.py:22-28)
    loaded = app.load_stn(args, ckpt, load_classifier=True, device=cuda)
    return app, args, loaded[0], loaded[1]


@pytest.mark.parametrize('case', load_golden('applications'), ids=lambda c: f"heads{c['meta']['num_heads']}")
def test_propagate_frames_golden(case, cuda):
    app, args, t, classifier = build(case, cuda)
    heads = case['meta']['num_heads']
    frames = T(case['frames'], cuda)
    pts_norm = T(case['points_norm'], cuda)
    assert (classifier is None) == (heads == 1) and not t.training
    if heads == 1:
        res = app.propagate_frames(args, t, None, frames, pts_norm, overlay=False)
        points, flips, clusters, cong = res['points'], res['flip_indices'], res['active_clusters'], res['congealed']
    else:
        outs = [app.propagate_frames(args, t, classifier, frames[i:i + 1], [pts_norm[k:k + 1] for k in range(heads)],
                                     overlay=False) for i in range(frames.size(0))]
        points = torch.cat([o['points'] for o in outs])
        flips = torch.cat([o['flip_indices'].reshape(-1) for o in outs])
        clusters = torch.cat([o['active_clusters'].reshape(-1) for o in outs])
        cong = torch.cat([o['congealed'] for o in outs])
    assert np.array_equal(flips.reshape(-1).cpu().numpy(), case['flip'])
    assert np.array_equal(clusters.cpu().numpy(), case['clusters'])
    close(points, case['points'], 2e-2, 1e-4)                                    # pixels of the uncropped frames
    assert list(cong.shape) == list(case['congealed'].shape)
    close(cong, case['congealed'], 2e-4, 1e-4)


def test_fixed_cluster_mode_and_overlay(cuda):
    (_, case) = load_golden('applications')
    app, args, t, classifier = build(case, cuda)
    frames = T(case['frames'], cuda)
    pts = [T(case['points_norm'][k:k + 1], cuda) for k in range(2)]
    n, p = frames.size(0), pts[0].size(1)
    colors = [torch.rand(1, p, 3, device=cuda) * 2 - 1 for _ in range(2)]
    alphas = [torch.rand(1, p, 1, device=cuda) for _ in range(2)]
    res = app.propagate_frames(args, t, classifier, frames, pts, colors=colors, alpha_channels=alphas, clusters=[1, 0],
                               sigma=1.3, opacity=0.7)
    assert res['points'].shape == (n, 2 * p, 2) and res['frame'].shape == frames.shape
    assert res['congealed'].shape == (n, 2, 3, args.real_size, args.real_size)
    # each half equals the single-cluster call
    for pos, c in enumerate((1, 0)):
        single = app.propagate_frames(args, t, classifier, frames, pts, clusters=[c], overlay=False)
        assert torch.equal(single['points'], res['points'][:, pos * p:(pos + 1) * p])
        flipped, flip_idx, policy, active = app.determine_flips(args, t, classifier, frames, cluster=c,
                                                                return_cluster_assignments=True)
        assert bool((active == c).all()) and policy.shape == (n, 2) and bool((policy.argmax(1) == c).all())
    # the overlay changed the frames where the labels landed and nowhere far away
    diff = (res['frame'] - frames).abs().amax(dim=1)
    assert float(diff.max()) > 0.05
    far = torch.ones_like(diff, dtype=torch.bool)
    for i in range(n):
        for x, y in res['points'][i].round().long().clamp(0, args.real_size - 1).tolist():
            far[i, max(y - 6, 0):y + 7, max(x - 6, 0):x + 7] = False
    assert float(diff[far].max()) < 1e-6
    # no_flip_inference: nothing is mirrored
    args.no_flip_inference = True
    same, flip_idx, policy = app.determine_flips(args, t, None, frames)
    assert same is frames and not bool(flip_idx.any()) and policy == 'cartesian'
