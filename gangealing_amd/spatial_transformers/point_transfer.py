"""Key-point transfer and object propagation on top of a trained Spatial Transformer: the
inference-side helpers of the reference's `spatial_transformer.py` (ComposedSTN :141-366, SpatialTransformer
:617-720) with the same names, arguments and conventions, written against this package's STN modules.  They are
thin torch glue around the kernels of the training path plus `splat2d` (csrc/splat2d.hip) in `propagate_object`.

Conventions (reference :617-629): pixel coordinates live in [0, H-1]; normalised coordinates are the
align_corners=False ones used by grid_sample, i.e. pixel p of an H-wide image maps to (p/(H-1) - 0.5) * 2 * (res-1)/res.
"""
import torch
import torch.nn.functional as F


def unravel_index(indices, shape):
    """Flat indices (*, N) -> coordinates (*, N, len(shape)), last dimension fastest first (reference :23-44:
    the result is ordered (x, y) for shape = (H, W))."""
    coords = []
    for dim in reversed(shape):
        coords.append(indices % dim)
        indices = torch.div(indices, dim, rounding_mode='floor')
    return torch.stack(coords, dim=-1)


def normalize(points, res, out_res):
    return points.div(out_res - 1).add(-0.5).mul(2).mul((res - 1) / res)


def unnormalize(points, res, out_res):
    return points.div((res - 1) / res).div(2).add(0.5).mul(out_res - 1)


def convert(points, current_res, target_res):
    return unnormalize(normalize(points, target_res, current_res), target_res, target_res)


def per_sample_total_variation(delta_flow):
    """total_variation_loss(delta_flow, reduce_batch=False) (losses/loss.py:4-12): Huber TV per batch element."""
    def huber_mean(a):
        return torch.where(a <= 1.0, 0.5 * a * a, a - 0.5).mean(dim=(1, 2, 3))
    dy = (delta_flow[:, :-1] - delta_flow[:, 1:]).abs()
    dx = (delta_flow[:, :, :-1] - delta_flow[:, :, 1:]).abs()
    return huber_mean(dx) + huber_mean(dy)


def _homogeneous(points):
    return torch.cat([points, points.new_ones(points.shape[0], points.shape[1], 1)], 2)


def _bottom_row(matrix):
    row = matrix.new_tensor([[[0.0, 0.0, 1.0]]]).repeat(matrix.shape[0], 1, 1)
    return torch.cat([matrix, row], 1)


def _lookup(grid, points):
    """Bilinear lookup of a sampling grid (N, H, W, 2) at normalised points (N, P, 2) -> (N, P, 2)."""
    out = F.grid_sample(grid.permute(0, 3, 1, 2), points.unsqueeze(2).float(), padding_mode='border',
                        align_corners=False)
    return out.squeeze(3).permute(0, 2, 1)


class SingleStnPointOps:
    """Mixed into SpatialTransformer (reference :617-720)."""

    normalize = staticmethod(normalize)
    unnormalize = staticmethod(unnormalize)
    convert = staticmethod(convert)

    def congeal_points(self, imgA, pointsA, normalize_input_points=True, unnormalize_output_points=False,
                       output_resolution=None, iters=1, input_img_for_sampling=None, return_full=False,
                       **stn_forward_kwargs):
        """Key points of imgA -> the congealed frame.  Similarity: the inverse matrix in closed form.  Flow: the
        STN only knows the congealed->A map, so the forward map is approximated by the nearest grid node."""
        assert imgA.size(0) == pointsA.size(0)
        n, num_points = imgA.size(0), pointsA.size(1)
        source_res = imgA.size(-1) if input_img_for_sampling is None else input_img_for_sampling.size(-1)
        outA, gridA, fm = self.forward(imgA, return_warp=True, return_flow=True, output_resolution=output_resolution,
                                       input_img_for_sampling=input_img_for_sampling, iters=iters, **stn_forward_kwargs)
        if normalize_input_points:
            pointsA = normalize(pointsA, source_res, source_res)
        if not self.is_flow:
            to_congealed = torch.inverse(_bottom_row(fm)).permute(0, 2, 1)
            points_congealed = (_homogeneous(pointsA) @ to_congealed)[..., [0, 1]]
            if unnormalize_output_points:
                points_congealed = unnormalize(points_congealed, source_res, source_res)
        else:
            assert fm.size(-1) == 2
            grid = fm + self.identity_flow                                   # (N, H, W, 2)
            h, w = grid.shape[1], grid.shape[2]
            nodes = grid.reshape(n, h * w, 1, 2)
            pts = pointsA.reshape(n, 1, num_points, 2)
            # ||x - y||^2 = ||x||^2 + ||y||^2 - 2<x, y>, as the reference evaluates it
            dist = pts.pow(2).sum(-1) + nodes.pow(2).sum(-1) - 2 * (nodes * pts).sum(-1)     # (N, H*W, P)
            points_congealed = unravel_index(dist.argmin(dim=1), (h, w))
        if return_full:
            return outA, fm, points_congealed
        return points_congealed

    def uncongeal_points(self, imgB, points_congealed, unnormalize_output_points=True, normalize_input_points=False,
                         output_resolution=None, iters=1, input_img_for_sampling=None, **stn_forward_kwargs):
        """Key points of the congealed frame -> imgB (the direction the STN's reverse sampling provides)."""
        assert imgB.size(0) == points_congealed.size(0)
        source_res = imgB.size(-1) if input_img_for_sampling is None else input_img_for_sampling.size(-1)
        _, gridB, fm = self.forward(imgB, return_warp=True, return_flow=True, output_resolution=output_resolution,
                                    iters=iters, input_img_for_sampling=input_img_for_sampling, **stn_forward_kwargs)
        if normalize_input_points:
            points_congealed = normalize(points_congealed, source_res, imgB.size(-1))
        if not self.is_flow:
            pointsB = (_homogeneous(points_congealed) @ _bottom_row(fm).permute(0, 2, 1))[..., [0, 1]]
        else:
            assert gridB.size(-1) == 2
            pointsB = _lookup(gridB, points_congealed)
        if unnormalize_output_points:
            pointsB = unnormalize(pointsB, imgB.size(-1), source_res)
        return pointsB

    def transfer_points(self, imgA, imgB, pointsA, output_resolution=None, iters=1, **stn_forward_kwargs):
        assert imgA.size(0) == imgB.size(0) == pointsA.size(0)
        mid = self.congeal_points(imgA, pointsA, output_resolution=output_resolution, iters=iters, **stn_forward_kwargs)
        return self.uncongeal_points(imgB, mid, output_resolution=output_resolution, normalize_input_points=False,
                                     iters=iters, **stn_forward_kwargs)


class ComposedStnPointOps:
    """Mixed into ComposedSTN (reference :141-376)."""

    def uncongeal_points(self, imgB, points_congealed, output_resolution=None, iters=1, unnormalize_output_points=True,
                         normalize_input_points=False, return_congealed_img=False, **stn_forward_kwargs):
        assert imgB.size(0) == points_congealed.size(0)
        if normalize_input_points:
            points_congealed = normalize(points_congealed, imgB.size(-1), self.stn_in_size)
        congealed_img, gridB = self.forward(imgB, return_warp=True, output_resolution=output_resolution, iters=iters,
                                            **stn_forward_kwargs)
        pointsB = _lookup(gridB, points_congealed)
        if unnormalize_output_points:
            pointsB = unnormalize(pointsB, imgB.size(-1), imgB.size(-1))
        return (pointsB, congealed_img) if return_congealed_img else pointsB

    def congeal_points(self, imgA, pointsA, output_resolution=None, iters=1, normalize_input_points=True,
                       unnormalize_output_points=False, return_full=False, **stn_forward_kwargs):
        assert imgA.size(0) == pointsA.size(0)
        out, warp, pts = imgA, None, pointsA
        last = self.N_minus_1
        for i, stn in enumerate(self.stns):
            out, warp, pts = stn.congeal_points(
                out, pts, normalize_input_points=normalize_input_points if i == 0 else True,
                unnormalize_output_points=unnormalize_output_points if i == last else True,
                iters=iters if i == 0 else 1, output_resolution=output_resolution if i == last else self.stn_in_size,
                base_warp=warp, input_img_for_sampling=imgA, return_full=True, **stn_forward_kwargs)
        return (out, warp, pts) if return_full else pts

    def transfer_points(self, imgA, imgB, pointsA, output_resolution=None, iters=1, congeal_kwargs={},
                        uncongeal_kwargs={}, **stn_forward_kwargs):
        assert imgA.size(0) == imgB.size(0) == pointsA.size(0)
        mid = self.congeal_points(imgA, pointsA, output_resolution=output_resolution, normalize_input_points=True,
                                  iters=iters, **congeal_kwargs, **stn_forward_kwargs)
        return self.uncongeal_points(imgB, mid, output_resolution=output_resolution, normalize_input_points=True,
                                     unnormalize_output_points=True, iters=iters, **uncongeal_kwargs,
                                     **stn_forward_kwargs)

    def forward_with_flip(self, input_img, return_flow=False, return_warp=False, return_inputs=False,
                          return_flip_indices=False, **stn_forward_kwargs):
        """Run the image and its mirror image; keep, per sample, whichever gives the smoother residual flow."""
        mirrored = input_img.flip(3)
        out, warp, flow = self.forward(input_img, return_warp=True, return_flow=True, **stn_forward_kwargs)
        outF, warpF, flowF = self.forward(mirrored, return_warp=True, return_flow=True, **stn_forward_kwargs)
        rough = torch.stack([per_sample_total_variation(flow), per_sample_total_variation(flowF)], 0)
        use_mirror = rough.argmin(dim=0).view(-1, 1, 1, 1).bool()
        result = [torch.where(use_mirror, outF, out)]
        if return_warp:
            warpF = warpF.clone()
            warpF[..., 0] = -warpF[..., 0]
            result.append(torch.where(use_mirror, warpF, warp))
        if return_flow:
            result.append(torch.where(use_mirror, flowF, flow))
        if return_inputs:
            result.append(torch.where(use_mirror, mirrored, input_img))
        if return_flip_indices:
            result.append(use_mirror)
        return result[0] if len(result) == 1 else result

    def match_flows(self, imgA, imgB, pointsA, pointsB=None, permutation=None, **stn_forward_kwargs):
        """Mirror imgA and / or imgB (per pair) so that their residual flows are jointly smoothest, and mirror the
        x coordinate (and, via `permutation`, the left/right labels) of the key points accordingly.
        pick: 0 none, 1 A mirrored, 2 B mirrored, 3 both (reference :242-295)."""
        a_m, b_m = imgA.flip(3), imgB.flip(3)
        _, flows = self.forward(torch.cat([imgA, imgB, a_m, b_m], 0), return_flow=True, **stn_forward_kwargs)
        tv_a, tv_b, tv_am, tv_bm = (per_sample_total_variation(f) for f in flows.chunk(4, dim=0))
        pick = torch.stack([tv_a + tv_b, tv_am + tv_b, tv_a + tv_bm, tv_am + tv_bm], 0).argmin(dim=0)
        pick = pick.view(imgA.size(0), 1, 1, 1)
        keep_a, keep_b = pick % 2 == 0, pick <= 1
        imgA = torch.where(keep_a, imgA, a_m)
        imgB = torch.where(keep_b, imgB, b_m)
        pointsA = pointsA.clone()
        pointsA[:, :, 0] = torch.where(keep_a.view(-1, 1), pointsA[:, :, 0], imgA.size(-1) - 1 - pointsA[:, :, 0])
        if permutation is not None:
            pointsA = torch.where(keep_a.view(-1, 1, 1), pointsA, pointsA[:, permutation])
        if pointsB is None:
            return imgA, imgB, pointsA, pick
        pointsB = pointsB.clone()
        pointsB[:, :, 0] = torch.where(keep_b.view(-1, 1), pointsB[:, :, 0], imgB.size(-1) - 1 - pointsB[:, :, 0])
        if permutation is not None:      # (the reference permutes pointsA here as well, :291-292)
            pointsA = torch.where(keep_b.view(-1, 1, 1), pointsA, pointsA[:, permutation])
        return imgA, imgB, pointsA, pointsB, pick

    def propagate_object(self, congealed_object_points, congealed_object_values, congealed_mask_values, target_image,
                         sigma, cluster_classifier=None, cluster=None, mem_efficient=False, **uncongeal_kwargs):
        """Paint an object defined in the congealed frame (points + RGB values + mask values) into the frame of
        `target_image`: move the points with `uncongeal_points`, drop those that leave the image, splat the rest with a
        Gaussian footprint (splat2d) and undo the cluster classifier's mirroring.  -> (object (N,C,H,W), mask (N,1,H,W))"""
        from ..splat2d_cuda import splat2d
        device = congealed_object_points.device
        n = congealed_object_points.size(0)
        assert n == congealed_mask_values.size(0) == target_image.size(0) == sigma.size(0), \
            'all tensor inputs should have the same batch size'
        size = target_image.size(-1)
        assert size == target_image.size(-2), 'square images only'
        assert congealed_object_points.dim() == congealed_mask_values.dim() == 3
        if self.num_heads == 1:
            policy = 'cartesian'
            flip = torch.zeros(n, device=device, dtype=torch.bool)
        else:
            assert cluster_classifier is not None, 'a cluster_classifier is required for clustering models'
            policy = torch.eye(self.num_heads, device=device)[cluster].unsqueeze(0).repeat(n, 1)
            flip = cluster_classifier.run_flip_target(target_image, cluster)
            flip = flip[1] if isinstance(flip, (tuple, list)) else flip
        flip = flip.view(n, 1, 1, 1)
        moved = self.uncongeal_points(target_image, congealed_object_points, normalize_input_points=False,
                                      unnormalize_output_points=True, warp_policy=policy, **uncongeal_kwargs)
        nearest = moved.round()
        inside = (nearest[..., 0] >= 0) & (nearest[..., 1] >= 0) & (nearest[..., 0] < size) & (nearest[..., 1] < size)
        keep = [torch.where(row)[0] for row in inside]
        counts = [k.numel() for k in keep]
        canvas = torch.zeros_like(target_image)
        if counts == [counts[0]] * n and not mem_efficient:        # same number of visible points: one batched splat
            idx = torch.stack(keep).unsqueeze(2)
            pts = moved.gather(1, idx.repeat(1, 1, 2))
            vals = congealed_object_values.gather(1, idx.repeat(1, 1, congealed_object_values.size(2)))
            mvals = congealed_mask_values.gather(1, idx)
            obj = splat2d(canvas, pts, vals, sigma, False)
            mask = splat2d(canvas[:, :1], pts, mvals, sigma, True)
        else:
            objs, masks = [], []
            for i in range(n):
                pts = moved[i:i + 1, keep[i]]
                objs.append(splat2d(canvas[:1], pts, congealed_object_values[i:i + 1, keep[i]], sigma[i:i + 1], False))
                masks.append(splat2d(canvas[:1, :1], pts, congealed_mask_values[i:i + 1, keep[i]], sigma[i:i + 1], True))
            obj, mask = torch.cat(objs, 0), torch.cat(masks, 0)
        return torch.where(flip, obj.flip(3), obj), torch.where(flip, mask.flip(3), mask)

    def load_single_state_dict(self, state_dict, index, strict=True):
        return self.stns[index].load_state_dict(state_dict, strict)

    def load_several_state_dicts(self, state_dicts, indices, strict=True):
        assert len(state_dicts) == len(indices)
        for sd, index in zip(state_dicts, indices):
            self.load_single_state_dict(sd, index, strict)
