"""splat_points: overlay (propagated) points on images - the GPU compute of the mixed-reality / propagation
applications (utils/vis_tools/helpers.py:134-194): two splat2d launches (colours; soft-normalised alpha) and an alpha
composite.  The Plotly colour-scale lookup and the Laplacian-pyramid blenders of the reference are presentation
options outside the hot path: pass `colors` explicitly and use blend_alg='alpha'."""
import torch

from .functional import splat2d


@torch.inference_mode()
def splat_points(images, points, sigma, opacity, colorscale=None, colors=None, alpha_channel=None, blend_alg='alpha'):
    """images (N, C, H, W) in [-1, 1]; points (N, P, 2) or (N, K, P, 2) pixel coordinates (x, y); sigma float or (N,);
    colors (N, P, C) (or (N, K*P, C)); alpha_channel (N, P, 1) or None (= opaque).  -> (N, C, H, W)."""
    assert images.dim() == 4
    assert points.dim() == 3 or points.dim() == 4
    n = images.size(0)
    if points.dim() == 4:
        points = points.reshape(points.size(0), points.size(1) * points.size(2), 2)
    if colors is None:
        raise NotImplementedError('splat_points: pass `colors` (N, P, C); the Plotly colour-scale lookup of the '
                                  'reference (helpers.py:125-132) is not part of this package')
    if blend_alg != 'alpha':
        raise NotImplementedError(f'splat_points: blend_alg={blend_alg!r}; only alpha compositing (helpers.py:185-186)')
    dev = images.device
    if alpha_channel is None:
        alpha_channel = torch.ones(n, points.size(1), 1, device=dev)
    if isinstance(sigma, (float, int)):
        sigma = torch.tensor(sigma, device=dev, dtype=torch.float).view(1).repeat(n)
    blank_img = torch.zeros(n, images.size(1), images.size(2), images.size(3), device=dev)
    blank_mask = torch.zeros(n, 1, images.size(2), images.size(3), device=dev)
    obj = splat2d(blank_img, points, colors, sigma, False)
    mask = splat2d(blank_mask, points, alpha_channel, sigma, True) * opacity
    return mask * obj + (1 - mask) * images
