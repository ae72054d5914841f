"""Training objective of GANgealing (models/losses/loss.py:4-92) on the HIP operators, plus a
VGG16-topology perceptual loss (models/losses/lpips.py:181-223 with lpips=False, i.e. the
default --loss_fn vgg_ssl, :13-17) whose convolutions run on the MFMA implicit-GEMM kernel."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .op import conv_mfma
from .spatial_transformers.flow_ops import flow_losses


def total_variation_loss(delta_flow, reduce_batch=True):
    if not reduce_batch:
        raise NotImplementedError('per-sample TV is only used by visualisation code')
    return flow_losses(delta_flow)[0]


def flow_identity_loss(delta_flow):
    return flow_losses(delta_flow)[1]


def sample_gan_supervised_pairs(generator, ll, resize_fake2stn, psi, batch, dim_latent, freeze_ll, device, z=None):
    """(unaligned G(w), aligned target G(mix(w, c))) - loss.py:21-29.  G pass #1 needs no graph."""
    if z is None:
        z = torch.randn(batch, dim_latent, device=device)
    with torch.no_grad():
        unaligned_in, w_noise = generator([z], noise=None, return_latents=True)
    with torch.set_grad_enabled(not freeze_ll):
        w_aligned = ll([w_noise[:, 0, :]], psi=psi)
        aligned_target, _ = generator(w_aligned, input_is_latent=True, noise=None,
                                      grad_latents=getattr(ll, 'inject_index', None))
        aligned_target = resize_fake2stn(aligned_target)
    return unaligned_in, aligned_target


def gangealing_loss(generator, stn, ll, loss_fn, resize_fake2stn, psi, batch, dim_latent, freeze_ll, device,
                    sample_from_full_res=False, **stn_kwargs):
    unaligned_in, aligned_target = sample_gan_supervised_pairs(generator, ll, resize_fake2stn, psi, batch, dim_latent,
                                                               freeze_ll, device)
    sampling_src = unaligned_in if sample_from_full_res else None
    aligned_pred, delta_flow = stn(resize_fake2stn(unaligned_in), return_flow=True,
                                   input_img_for_sampling=sampling_src, **stn_kwargs)
    return loss_fn(aligned_pred, aligned_target).mean(), delta_flow


def assign_fake_images_to_clusters(generator, stn, ll, loss_fn, resize_fake2stn, psi, batch, dim_latent, freeze_ll,
                                   num_heads, flips, device, sample_from_full_res=True, z=None, **stn_kwargs):
    unaligned_in, aligned_target = sample_gan_supervised_pairs(generator, ll, resize_fake2stn, psi, batch, dim_latent,
                                                               freeze_ll, device, z)
    if flips:
        unaligned_in = torch.cat([unaligned_in, unaligned_in.flip(3)], 0)
        aligned_target = aligned_target.repeat(2, 1, 1, 1)
        loss_size = (2, batch, num_heads)
    else:
        loss_size = (batch, num_heads)
    sampling_src = unaligned_in if sample_from_full_res else None
    resized = resize_fake2stn(unaligned_in)
    aligned_pred, delta_flow = stn(resized, return_flow=True, input_img_for_sampling=sampling_src, **stn_kwargs)
    ploss = loss_fn(aligned_pred, aligned_target).view(*loss_size)
    collapsed = ploss.permute(1, 0, 2).reshape(batch, 2 * num_heads) if flips else ploss
    return collapsed.min(dim=1), aligned_pred, delta_flow, unaligned_in, resized, collapsed


def gangealing_cluster_loss(generator, stn, ll, loss_fn, resize_fake2stn, psi, batch, dim_latent, freeze_ll, num_heads,
                            flips, device, sample_from_full_res=True, **stn_kwargs):
    assignments, _, delta_flow, _, _, _ = assign_fake_images_to_clusters(
        generator, stn, ll, loss_fn, resize_fake2stn, psi, batch, dim_latent, freeze_ll, num_heads, flips, device,
        sample_from_full_res, **stn_kwargs)
    hw2 = delta_flow.size()[1:]
    if flips:
        delta_flow = delta_flow.view(2, batch, num_heads, *hw2).permute(1, 0, 2, 3, 4, 5).reshape(batch, 2 * num_heads, *hw2)
    else:
        delta_flow = delta_flow.view(batch, num_heads, *hw2)
    delta_flow = delta_flow[torch.arange(batch, device=delta_flow.device), assignments.indices]
    return assignments.values.mean(), delta_flow.contiguous()


class _LpipsTail(torch.autograd.Function):
    """mean_pixels sum_c lin[c] * (normalize(f0) - normalize(f1))^2 for one feature tap, fused
    (csrc/lpips.hip; lpips.py:26-28,190-199).  feats: (2N, C, H, W), first half image 0."""

    @staticmethod
    def forward(ctx, feats, lin, eps):
        feats = feats.contiguous()
        n2, c, h, w = feats.shape
        n = n2 // 2
        out = torch.empty(n, dtype=torch.float32, device=feats.device)
        _lib.call('gg_lpips_tail_fwd_f32', out, feats, lin, n, c, h * w, eps)
        ctx.save_for_backward(feats, lin if lin is not None else feats.new_empty(0))
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, grad_out):
        feats, lin = ctx.saved_tensors
        n2, c, h, w = feats.shape
        dfeats = torch.empty_like(feats)
        _lib.call('gg_lpips_tail_bwd_f32', dfeats, feats, lin if lin.numel() else None, grad_out.contiguous().float(),
                  n2 // 2, c, h * w, ctx.eps)
        return dfeats, None, None


def lpips_tail(feats, lin=None, eps=1e-10):
    if lin is not None:
        lin = lin.reshape(-1).contiguous().float()
    return _LpipsTail.apply(feats, lin, eps)


VGG16_CFG = [(64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512)]


class VGGPerceptualLoss(nn.Module):
    """LPIPS-style distance on VGG16 features (relu1_2 ... relu5_3), lpips=False branch:
    unit-normalise channels, squared difference, channel sum, spatial mean, summed over the 5 taps,
    divided by 18 (lpips.py:13-17,26-28,181-206).  Weights are random unless a state_dict in
    torchvision's vgg16().features layout is loaded (no checkpoint is reachable offline)."""

    def __init__(self, seed=0):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.convs = nn.ModuleList()
        cin = 3
        for stage in VGG16_CFG:
            for cout in stage:
                conv = nn.Conv2d(cin, cout, 3, padding=1)
                with torch.no_grad():
                    conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (cin * 9)) ** 0.5)
                    conv.bias.zero_()
                self.convs.append(conv)
                cin = cout
        self.register_buffer('shift', torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer('scale', torch.tensor([.458, .448, .450])[None, :, None, None])
        self.requires_grad_(False)

    def features(self, x):
        x = (x - self.shift) / self.scale
        feats, i = [], 0
        for si, stage in enumerate(VGG16_CFG):
            if si > 0:
                x = F.max_pool2d(x, 2, 2)
            for _ in stage:
                conv = self.convs[i]
                if x.shape[1] % 32 == 0:     # conv + bias + ReLU in one kernel (alpha 0, gain 1)
                    x = conv_mfma.conv3x3_bias_act(x, conv.weight, conv.bias, 0.0, 1.0)
                else:                        # 3-channel stem: fp32 kernel + separate ReLU
                    x = F.relu(conv_mfma.conv2d(x, conv.weight, conv.bias, stride=1, padding=1))
                i += 1
            feats.append(x)
        return feats

    def forward(self, in0, in1):
        n = in0.shape[0]
        feats = self.features(torch.cat([in0, in1], 0))       # one batched pass for both images
        val = 0
        for f in feats:
            if f.dtype == torch.float32 and 'lpips_tail' not in conv_mfma.DISABLED:
                val = val + lpips_tail(f).view(n, 1, 1, 1)     # normalize -> diff^2 -> channel sum -> mean, one kernel
                continue
            f = f / (torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True)) + 1e-10)
            d = (f[:n] - f[n:]) ** 2
            val = val + d.sum(dim=1, keepdim=True).mean(dim=(2, 3), keepdim=True)
        return val / 18.0


def get_perceptual_loss(loss_fn, device):
    if loss_fn not in ('vgg_ssl', 'lpips'):
        raise NotImplementedError(loss_fn)
    return VGGPerceptualLoss().to(device)
