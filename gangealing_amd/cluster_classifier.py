"""Cluster classifier of the clustering variants of GANgealing: same names, constructor and
state_dict keys as the reference `models/cluster_classifier.py:8-48` (ResnetClassifier) plus the training iteration
of `train_cluster_classifier.py:78-101`.  The trunk is the STN trunk (ConvLayer / ResBlock on the MFMA
convolution kernels), so nothing here needs a kernel of its own.

Given an image the classifier predicts (1) which learned cluster it belongs to and (2), when trained with
--flips, whether it should be mirrored before its cluster's STN sees it: head h < num_heads/2 = cluster h
un-flipped, head h + num_heads/2 = the same cluster flipped.
"""
import math

import torch
from torch import nn

from .spatial_transformers.antialiased_sampling import BilinearDownsample
from .stylegan2.networks import ConvLayer, EqualLinear, ResBlock


def accuracy(predictions, gt_probabilities, k=1):
    """"Reverse" top-k accuracy (models/__init__.py:37-43): is the classifier's arg-max among the k best classes
    of the ground-truth scores?"""
    choice = predictions.argmax(dim=1, keepdim=True)
    best = gt_probabilities.topk(k=k, dim=1).indices
    return (choice == best).any(dim=1).float().mean()


class ResnetClassifier(nn.Module):
    def __init__(self, size, channel_multiplier=2, blur_kernel=(1, 3, 3, 1), num_heads=1, supersize=None):
        super().__init__()
        self.stn_in_size = size
        self.num_heads = num_heads
        if supersize is not None:
            self.input_downsample = BilinearDownsample(supersize // size, 3)
        # channel table of cluster_classifier.py:16-26
        chan = {4: 512, 8: 512, 16: 512, 32: 512, 64: int(256 * channel_multiplier), 128: int(128 * channel_multiplier),
                256: int(64 * channel_multiplier), 512: int(32 * channel_multiplier), 1024: int(16 * channel_multiplier)}
        log_size = int(math.log2(size))
        width = chan[size]
        blocks = [ConvLayer(3, width, 1)]
        for i in range(log_size, 2, -1):
            nxt = chan[2 ** (i - 1)]
            blocks.append(ResBlock(width, nxt, blur_kernel))
            width = nxt
        self.convs = nn.Sequential(*blocks)
        self.final_conv = ConvLayer(width, chan[4], 3)
        self.to_logits = EqualLinear(chan[4] * 4 * 4, num_heads, activation='fused_lrelu')

    def forward(self, input):
        if input.size(-1) > self.stn_in_size:
            input = self.input_downsample(input)
        feats = self.final_conv(self.convs(input))
        return self.to_logits(feats.reshape(feats.size(0), -1))

    # ---- inference helpers (cluster_classifier.py:50-98) ----------------------------------------
    def assign(self, input, ignore_flips=False):
        cls = self.forward(input).argmax(dim=1)
        return cls % (self.num_heads // 2) if ignore_flips else cls

    @staticmethod
    def _mirror_where(mask, images, width_dim=3):
        return torch.where(mask, images.flip(width_dim), images)

    def run(self, input, target_cluster, return_flip_indices=False):
        half = self.num_heads // 2
        logits = self.forward(input)
        cls = logits.argmax(dim=1)
        (keep,) = torch.where((cls % half) == target_cluster)
        kept = input[keep]
        flip = (cls[keep] >= half).reshape(-1, 1, 1, 1)
        kept = self._mirror_where(flip, kept)
        if return_flip_indices:
            return kept, logits[keep], flip, keep
        return kept, logits[keep]

    def run_flip(self, input):
        half = self.num_heads // 2
        logits = self.forward(input)
        cls = logits.argmax(dim=1)
        flip = cls >= half
        return self._mirror_where(flip.reshape(-1, 1, 1, 1), input), logits, cls, flip

    def run_flip_target(self, input, target_cluster):
        half = self.num_heads // 2
        pair = self.forward(input)[:, [target_cluster, target_cluster + half]]
        flip = pair.argmax(dim=1) == 1
        return self._mirror_where(flip.reshape(-1, 1, 1, 1), input), flip

    def run_flip_cartesian(self, input):
        half = self.num_heads // 2
        n = input.size(0)
        flip = self.forward(input).view(n, 2, half).argmax(dim=1) == 1            # (N, half)
        tiled = input.unsqueeze(1).repeat(1, half, 1, 1, 1)
        tiled = self._mirror_where(flip.reshape(n, half, 1, 1, 1), tiled, width_dim=4)
        tiled = tiled.view(n * half, *input.shape[1:])
        policy = torch.eye(half, device=input.device).repeat(n, 1)
        return tiled, policy

    def load_state_dict(self, state_dict, strict=True):
        # the bilinear kernels are buffers derived from the constructor arguments
        skip = {'input_downsample.kernel_horz', 'input_downsample.kernel_vert'}
        from .spatial_transformers.spatial_transformer import load_filtered_state_dict
        return load_filtered_state_dict(self, state_dict, strict, skip)


def cluster_classifier_step(classifier, generator, t_ema, ll, loss_fn, resize_fake2stn, batch, dim_latent, num_heads,
                            flips, device, sample_from_full_res=True, z=None, **stn_kwargs):
    """Loss of one classifier-training iteration (train_cluster_classifier.py:78-92): label every fake image with
    the (cluster, flip) whose STN aligns it best - no gradient through image formation or assignment - and fit the
    classifier to those labels.  Returns (cross entropy, {metrics})."""
    from .losses import assign_fake_images_to_clusters
    with torch.no_grad():
        assignments, _, _, _, resized, distance = assign_fake_images_to_clusters(
            generator, t_ema, ll, loss_fn, resize_fake2stn, 0.0, batch, dim_latent, True, num_heads, flips, device,
            sample_from_full_res=sample_from_full_res, z=z, **stn_kwargs)
    logits = classifier(resized[:batch])
    xent = nn.functional.cross_entropy(logits, assignments.indices)
    total = num_heads * (1 + int(bool(flips)))
    metrics = {'cross_entropy': xent.detach(), 'acc@1': accuracy(logits, -distance), 'acc@2': accuracy(logits, -distance, k=2),
               'assignments': torch.bincount(assignments.indices, minlength=total).float() / batch,
               'predicted': torch.bincount(logits.argmax(dim=1), minlength=total).float() / batch}
    return xent, metrics
