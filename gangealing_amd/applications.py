"""Inference-side glue of the reference's applications package on the HIP operators: loading a trained STN
(applications/__init__.py:30-57), deciding per image whether to mirror it before congealing (:60-94) and the per-batch
body of the augmented-reality loop that propagates dense labels from the congealed frame to video frames
(applications/mixed_reality.py:147-218).  No CLI, no video / image I/O (out of scope): callers
hand over tensors.

`args` is any object with the attributes of base_eval_argparse (applications/__init__.py:7-27): transform, flow_size,
stn_channel_multiplier, num_heads, real_size (or crop_size), iters, padding_mode, no_flip_inference."""
import torch

from .cluster_classifier import ResnetClassifier
from .spatial_transformers.spatial_transformer import get_stn
from .splat2d_cuda.overlay import splat_points


def load_stn(args, ckpt, load_classifier=False, device='cuda'):
    """ckpt: a checkpoint dict in the reference's layout (train.py:22-28: key 't_ema', optionally 'classifier') or a
    path to one.  The reference resolves names of its published models through utils/download.find_model; there is no
    network here, so only files / dicts are taken.  -> t_ema, or (t_ema, classifier | None)."""
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt, map_location='cpu')
    supersize = getattr(args, 'crop_size', None) or args.real_size
    t_ema = get_stn(args.transform, flow_size=args.flow_size, supersize=supersize,
                    channel_multiplier=args.stn_channel_multiplier, num_heads=args.num_heads).to(device)
    t_ema.load_state_dict(ckpt['t_ema'])
    t_ema.eval()
    if not load_classifier:
        return t_ema
    classifier = None
    if 'classifier' in ckpt:
        classifier = ResnetClassifier(args.flow_size, channel_multiplier=args.stn_channel_multiplier,
                                      num_heads=2 * args.num_heads, supersize=supersize).to(device)
        classifier.load_state_dict(ckpt['classifier'])
        classifier.eval()
    return t_ema, classifier


def determine_flips(args, t, classifier, input_imgs, cluster=None, return_cluster_assignments=False):
    """Which images should be mirrored before the STN sees them (applications/__init__.py:60-94): with a cluster
    classifier its prediction decides (and selects the head: `warp_policy` is the one-hot head assignment); otherwise
    both img and flip(img) go through the STN and the smoother residual flow wins (forward_with_flip), unless
    args.no_flip_inference.  -> (possibly mirrored images, flip mask (N, 1, 1, 1), warp_policy[, cluster indices])."""
    n, dev = input_imgs.size(0), input_imgs.device
    if classifier is not None:
        if cluster is None:
            data_flipped, _, clusters, flip_indices = classifier.run_flip(input_imgs)
            clusters = clusters % args.num_heads
        else:
            data_flipped, flip_indices = classifier.run_flip_target(input_imgs, cluster)
            clusters = torch.full((n,), int(cluster), dtype=torch.long, device=dev)
        warp_policy = torch.eye(args.num_heads, device=dev)[clusters]
    elif not args.no_flip_inference:
        _, data_flipped, flip_indices = t.forward_with_flip(input_imgs, return_inputs=True, return_flip_indices=True,
                                                            padding_mode=args.padding_mode, iters=args.iters)
        warp_policy = 'cartesian'
        clusters = torch.zeros(n, dtype=torch.long, device=dev)
    else:
        data_flipped = input_imgs
        flip_indices = torch.zeros(n, 1, 1, 1, device=dev, dtype=torch.bool)
        warp_policy = 'cartesian'
        clusters = torch.zeros(n, dtype=torch.long, device=dev)
    if return_cluster_assignments:
        return data_flipped, flip_indices, warp_policy, clusters
    return data_flipped, flip_indices, warp_policy


def center_crop_square(frames):
    """(N, C, H, W) -> (square center crop, (y_start, x_start)) - nchw_center_crop of the reference's video loader."""
    h, w = frames.shape[-2:]
    s = min(h, w)
    y0, x0 = (h - s) // 2, (w - s) // 2
    return frames[..., y0:y0 + s, x0:x0 + s], (y0, x0)


def propagate_frames(args, t, classifier, frames, points, colors=None, alpha_channels=None, clusters=None,
                     overlay=True, sigma=1.3, opacity=0.7, blend_alg='alpha'):
    """One batch of the mixed-reality loop (mixed_reality.py:147-218): propagate congealed-frame points (already
    normalised to [-1, 1], shape (1 | N, P, 2); with a classifier: a list with one entry per cluster) to every frame.

      clusters None  -> 'unimodal' (no classifier) / 'predict_cluster' (classifier picks the active cluster; batch 1)
      clusters [c..] -> 'fixed_cluster': the labels of each listed cluster are propagated and concatenated

    -> dict(points (N, P', 2) in pixel coordinates of the UNCROPPED frames, flip_indices, active_clusters, frame
    (frames with the labels splatted on, when overlay and colors are given), congealed (STN(frames) at real_size))."""
    n = frames.size(0)
    original = frames
    crop = frames.size(2) != frames.size(3)
    y0 = x0 = 0
    if crop:
        frames, (y0, x0) = center_crop_square(frames)
        frames = frames.contiguous()
    clustering = classifier is not None

    def one(cluster):
        flipped, flip_idx, policy, active = determine_flips(args, t, classifier, frames, cluster=cluster,
                                                            return_cluster_assignments=True)
        if clustering and cluster is None:
            assert n == 1, 'predict_cluster propagates one frame at a time (mixed_reality.py:200)'
            pts_in = points[int(active.item())]
        elif clustering:
            pts_in = points[cluster]
        else:
            pts_in = points
        pts_in = pts_in.expand(n, -1, -1) if pts_in.size(0) == 1 else pts_in
        out = t.uncongeal_points(flipped, pts_in, normalize_input_points=False, warp_policy=policy,
                                 padding_mode=args.padding_mode, iters=args.iters)
        # mirrored frames: mirror the propagated x coordinate back (mixed_reality.py:168-170)
        out[:, :, 0] = torch.where(flip_idx.view(-1, 1), args.real_size - 1 - out[:, :, 0], out[:, :, 0])
        return out, flipped, flip_idx, policy, active

    if clusters is None:
        pts, flipped, flip_idx, policy, active = one(None)
    else:
        parts = [one(c) for c in clusters]
        pts = torch.cat([p[0] for p in parts], 1)
        flipped, flip_idx, policy = parts[-1][1], parts[-1][2], parts[-1][3]
        active = torch.cat([p[4] for p in parts], 0)
    if crop:
        pts[:, :, 0] += x0
        pts[:, :, 1] += y0
    res = dict(points=pts, flip_indices=flip_idx, active_clusters=active)
    if overlay and colors is not None:
        if clustering and clusters is None:
            col, alp = colors[int(active.item())], alpha_channels[int(active.item())]
        elif clustering:
            col = torch.cat([colors[c] for c in clusters], 1)
            alp = torch.cat([alpha_channels[c] for c in clusters], 1)
        else:
            col, alp = colors, alpha_channels
        col = col.expand(n, -1, -1) if col.size(0) == 1 else col
        alp = alp.expand(n, -1, -1) if alp is not None and alp.size(0) == 1 else alp
        res['frame'] = splat_points(original, pts, sigma=sigma, opacity=opacity, colors=col, alpha_channel=alp,
                                    blend_alg=blend_alg)
    if clustering:
        flipped, policy = classifier.run_flip_cartesian(frames)
    congealed = t(flipped, output_resolution=args.real_size, warp_policy=policy, unfold=clustering,
                  padding_mode=args.padding_mode, iters=args.iters)
    res['congealed'] = congealed if clustering else congealed.unsqueeze(1)
    return res
