"""DirectionInterpolator: the learned target latent c (models/latent_learner.py:25-83).  Buffer and
parameter names (``directions``, ``lat_mean``, ``coefficients``) match the reference.  The
init-time PCA / k-means++ (sklearn, :8-22,85-123) is host-side preparation outside the hot path:
feed its result through assign_buffers()."""
import torch
import torch.nn as nn


class DirectionInterpolator(nn.Module):
    def __init__(self, pca_path, n_comps, inject_index, n_latent, num_heads=1, initializer=None, dim_latent=512):
        super().__init__()
        if pca_path is not None:
            import numpy as np
            with np.load(pca_path) as data:
                self.register_buffer('lat_mean', torch.from_numpy(data['lat_mean']))
                self.register_buffer('directions', torch.from_numpy(data['lat_comp'].squeeze(axis=1))[:n_comps])
        else:   # placeholders, overwritten by assign_buffers or a checkpoint
            self.register_buffer('directions', torch.randn(n_comps, dim_latent))
            self.register_buffer('lat_mean', torch.randn(1, dim_latent))
        if initializer is None:
            initializer = torch.zeros(num_heads, n_comps)
        self.coefficients = nn.Parameter(initializer.detach().clone())
        self.n_latent = n_latent
        self.inject_index = inject_index
        self.num_heads = num_heads

    def forward(self, styled_latent, psi=None, lat_mean=None, pca=None, unfold=False):
        if pca is not None:
            return self.assign_buffers(pca)
        return self.interpolate(styled_latent, psi, lat_mean, unfold)

    def interpolate(self, styled_latent, psi, lat_mean=None, unfold=False):
        """W+ codes of the aligned targets: the first `inject_index` slots carry
        lerp(lat_mean + coefficients @ directions, w, psi), the rest carry w (:56-70)."""
        assert len(styled_latent) == 1
        w = styled_latent[0]
        n = w.size(0)
        mean = self.lat_mean if lat_mean is None else lat_mean
        target = (mean + self.coefficients @ self.directions).repeat(n, 1)          # (N*K, D)
        w = w.repeat_interleave(self.num_heads, dim=0)
        head = target.lerp(w, psi).unsqueeze(1).repeat(1, self.inject_index, 1)
        tail = w.unsqueeze(1).repeat(1, self.n_latent - self.inject_index, 1)
        out = torch.cat([head, tail], dim=1)
        if unfold:
            out = out.reshape(n, self.num_heads, self.n_latent, -1)
        return [out]

    @torch.no_grad()
    def assign_buffers(self, pca):
        """pca: object with .pca.components_ / .pca.mean_ (reference PCA wrapper) or (directions, mean) tensors."""
        if isinstance(pca, (tuple, list)):
            comp, mean = pca
        else:
            comp = torch.from_numpy(pca.pca.components_).float()
            mean = torch.from_numpy(pca.pca.mean_[None]).float()
        dev = self.coefficients.device
        self.register_buffer('directions', comp.to(dev))
        self.register_buffer('lat_mean', mean.to(dev))

    def assign_coefficients(self, initializer):
        with torch.no_grad():
            self.coefficients.copy_(initializer)
