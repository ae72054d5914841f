"""Training objective of GANgealing (models/losses/loss.py:4-92) on the HIP operators, plus a
VGG16-topology perceptual loss (models/losses/lpips.py:181-223 with lpips=False, i.e. the
default --loss_fn vgg_ssl, :13-17) whose convolutions run on the MFMA implicit-GEMM kernel."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

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
    """(unaligned G(w), aligned target G(mix(w, c))) - loss.py:21-29.  G pass #1 needs no graph.

    Both synthesis passes run on the caller's stream.  (Rounds 2 - 5 could fork pass #1 onto a second HIP stream:
    +0.3 % at per-GPU batch 5, +0.6 % at 16 - and 5 % of repeated runs of one small configuration lost bitwise
    reproducibility with it, profiles/r05_d_two_stream_determinism.txt.  The state the two streams raced on was never
    located, so round 6 removed the fork instead of shipping it behind a switch.)"""
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
                  n2 // 2, c, h * w, ctx.eps, 0)
        return dfeats, None, None


def lpips_tail(feats, lin=None, eps=1e-10):
    if lin is not None:
        lin = lin.reshape(-1).contiguous().float()
    return _LpipsTail.apply(feats, lin, eps)


class _LpipsTap(torch.autograd.Function):
    """A feature tap of the trunk as ONE autograd node: returns (the features, passed on to the next stage, and this
    tap's distance).  In the backward the tail's gradient is ADDED by the tail kernel into the gradient that came
    back from the next stage, instead of being written to a second full-size tensor that autograd then adds."""

    @staticmethod
    def forward(ctx, feats, lin, eps):
        feats = feats.contiguous()
        n2, c, h, w = feats.shape
        out = torch.empty(n2 // 2, dtype=torch.float32, device=feats.device)
        _lib.call('gg_lpips_tail_fwd_f32', out, feats, lin, n2 // 2, c, h * w, eps)
        ctx.save_for_backward(feats, lin if lin is not None else feats.new_empty(0))
        ctx.eps = eps
        # the last tap's features feed nothing else: without this autograd would hand backward() a full-size zero
        # tensor for them (a memset plus a read-modify-write of the 512-channel map)
        ctx.set_materialize_grads(False)
        return feats, out

    @staticmethod
    def backward(ctx, g_feats, g_out):
        feats, lin = ctx.saved_tensors
        n2, c, h, w = feats.shape
        if g_out is None:
            return g_feats, None, None
        if g_feats is None:
            df, acc = torch.empty_like(feats), 0
        else:
            # Updated IN PLACE.  Invariant this relies on: the incoming gradient was produced for this node alone - it is
            # the data gradient written by the next stage's first convolution (a fresh tensor of _Conv3x3BiasAct /
            # _MaxPool2x2.backward).  No backward in this package hands one gradient tensor to two consumers that mutate
            # (conv_mfma._AddScale returns one tensor twice, to consumers that only read it).
            df, acc = g_feats.contiguous(), 1
        _lib.call('gg_lpips_tail_bwd_f32', df, feats, lin if lin.numel() else None, g_out.contiguous().float(),
                  n2 // 2, c, h * w, ctx.eps, acc)
        return df, None, None


def lpips_tap(feats, lin=None, eps=1e-10):
    """-> (features for the next stage, per-sample distance of this tap)."""
    if lin is not None:
        lin = lin.reshape(-1).contiguous().float()
    return _LpipsTap.apply(feats, lin, eps)


VGG16_CFG = [(64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512)]
# positions of the conv layers inside torchvision's vgg16().features, grouped by the reference's slices
# (lpips_backbones.py:109-118): slice1 = features[0:4], slice2 = [4:9], slice3 = [9:16], slice4 = [16:23], slice5 = [23:30]
_VGG16_SLICES = [(0, 4), (4, 9), (9, 16), (16, 23), (23, 30)]
_VGG16_FEATURES = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


def vgg16_feature_layers():
    """[(index in torchvision's vgg16().features, module)] for indices 0..29 (conv / ReLU / max-pool)."""
    layers, cin = [], 3
    for v in _VGG16_FEATURES:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return list(enumerate(layers))[:30]


class _MaxPool2x2(Function):
    """F.max_pool2d(x, 2, 2) of the VGG16 trunk on gg_maxpool2x2_*: the forward keeps one byte per output (which of the
    four inputs won, ATen's tie rule) instead of an int64 index, the backward writes dx in one pass."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        n, c, h, w = x.shape
        out = torch.empty((n, c, h // 2, w // 2), dtype=x.dtype, device=x.device)
        code = torch.empty((n, c, h // 2, w // 2), dtype=torch.uint8, device=x.device)
        _lib.call('gg_maxpool2x2_fwd_f32', out, code, x, n * c, h, w)
        ctx.save_for_backward(code)
        ctx.shape = (n, c, h, w)
        return out

    @staticmethod
    def backward(ctx, grad):
        (code,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        dx = torch.empty((n, c, h, w), dtype=grad.dtype, device=grad.device)
        _lib.call('gg_maxpool2x2_bwd_f32', dx, grad.contiguous(), code, n * c, h, w)
        return dx


def max_pool2x2(x):
    """2x2 / stride-2 max pooling; odd sizes (floor mode drops the last row / column) and non-fp32 tensors use ATen's
    operator on the same device (no host path: the HIP entry point raises for tensors that are not on the GPU)."""
    if (x.dtype != torch.float32 or x.dim() != 4 or x.shape[-1] % 2 or x.shape[-2] % 2
            or 'maxpool' in conv_mfma.DISABLED):          # GG_DISABLE=maxpool: A/B measurements
        return F.max_pool2d(x, 2, 2)
    return _MaxPool2x2.apply(x)


class vgg16(nn.Module):
    """VGG16 trunk with the module names of the reference wrapper (lpips_backbones.py:98-140: `slice1`..`slice5`,
    children named by their torchvision `features` index), so both torchvision-layout `features` state_dicts
    (`pretrained_weights`, e.g. simclr_vgg_phase150.pt: keys "0.weight", "2.weight", ...) and reference LPIPS
    checkpoints (keys "net.slice1.0.weight", ...) load.  Convolutions run on the MFMA kernels with bias + ReLU in
    the epilogue; torchvision itself is not needed."""

    def __init__(self, requires_grad=False, pretrained=True, pretrained_weights=None, seed=0):
        super().__init__()
        layers = vgg16_feature_layers()
        for si, (lo, hi) in enumerate(_VGG16_SLICES):
            seq = nn.Sequential()
            for idx, mod in layers[lo:hi]:
                seq.add_module(str(idx), mod)
            setattr(self, f'slice{si + 1}', seq)
        self.N_slices = 5
        self.weights_loaded = False
        gen = torch.Generator().manual_seed(seed)
        with torch.no_grad():                      # seeded He initialisation until real weights are loaded
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    fan_in = m.in_channels * 9
                    m.weight.copy_(torch.randn(m.weight.shape, generator=gen) * (2.0 / fan_in) ** 0.5)
                    # a small positive bias keeps some channels active at every pixel: with all-zero biases a random
                    # trunk produces pixels whose feature vector is (almost) exactly zero, where the perceptual
                    # distance's x / (|x| + 1e-10) has gradients of order 1e10 .. 1e20 (lpips.py:26-28) - enough to
                    # overflow Adam's second moment in float32 within a few dozen iterations of a synthetic run
                    m.bias.fill_(0.1)
        if pretrained_weights is not None:
            self.load_features_state_dict(torch.load(pretrained_weights, map_location='cpu'), strict=True)
        elif pretrained:
            raise FileNotFoundError('vgg16(pretrained=True) would need torchvision\'s ImageNet weights, which cannot be '
                                    'downloaded here; pass pretrained_weights=<features state_dict file> or pretrained=False')
        if not requires_grad:
            self.requires_grad_(False)

    def load_features_state_dict(self, sd, strict=True):
        """`sd`: state_dict of torchvision's vgg16().features ("<idx>.weight" / "<idx>.bias"; a "features." prefix
        is accepted)."""
        sd = {k[len('features.'):] if k.startswith('features.') else k: v for k, v in sd.items()}
        mapped, expected = {}, set()
        for si, (lo, hi) in enumerate(_VGG16_SLICES):
            for idx, mod in getattr(self, f'slice{si + 1}').named_children():
                if isinstance(mod, nn.Conv2d):
                    for leaf in ('weight', 'bias'):
                        expected.add(f'{idx}.{leaf}')
                        if f'{idx}.{leaf}' in sd:
                            mapped[f'slice{si + 1}.{idx}.{leaf}'] = sd[f'{idx}.{leaf}']
        unexpected = sorted(set(sd) - expected)
        missing = sorted(expected - set(sd))
        if strict and (unexpected or missing):
            raise RuntimeError(f'VGG16 features state_dict: missing {missing}, unexpected {unexpected}')
        result = self.load_state_dict(mapped, strict=False)
        self.weights_loaded = True
        return result

    def forward(self, x, tap=None, observe=None):
        """tap: optional callable (slice index, features) -> (features, value); when given the trunk returns the list
        of values instead of the list of feature maps (LPIPS' fused per-tap distance).
        observe: optional callable (torchvision `features` index of a convolution, its ReLU output) - diagnostics
        (tests/test_gpu_lpips_masks.py inspects / pins the branch decisions through it)."""
        if observe is not None and conv_mfma.ACT_OBSERVER is None:
            # a diagnostics hook may edit the outputs: the layers must keep no sign plane beside them (conv_mfma.ACT_OBSERVER)
            conv_mfma.ACT_OBSERVER = lambda site, y: None
            try:
                return self.forward(x, tap=tap, observe=observe)
            finally:
                conv_mfma.ACT_OBSERVER = None
        feats = []
        for si in range(5):
            for name, mod in getattr(self, f'slice{si + 1}').named_children():
                if isinstance(mod, nn.Conv2d):
                    if x.shape[1] % 32 == 0 or (x.shape[1] <= 4 and x.shape[-1] % 4 == 0 and
                                                'vgg_stem' not in conv_mfma.DISABLED):
                        # conv + bias + ReLU in one kernel (alpha 0, gain 1): the MFMA tiles, or - the 3-channel stem -
                        # the streaming few-input-channel kernel (exact fp32 products; csrc/conv_mfma.hip)
                        x = conv_mfma.conv3x3_bias_act(x, mod.weight, mod.bias, 0.0, 1.0)
                    else:                        # fp32 kernel + separate ReLU
                        x = F.relu(conv_mfma.conv2d(x, mod.weight, mod.bias, stride=1, padding=1))
                    if observe is not None:
                        observe(int(name), x)
                elif isinstance(mod, nn.MaxPool2d):
                    x = max_pool2x2(x)
                # nn.ReLU: applied in the convolution above
            if tap is not None:
                x, val = tap(si, x)
                feats.append(val)
            else:
                feats.append(x)
        return feats


class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer('shift', torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer('scale', torch.Tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, inp):
        return (inp - self.shift) / self.scale


class NetLinLayer(nn.Module):
    """1x1 convolution to one channel (lpips.py:235-245); evaluated inside the fused tail kernel."""

    def __init__(self, chn_in, chn_out=1, use_dropout=False):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class LPIPS(nn.Module):
    """Perceptual distance on VGG16 features (models/losses/lpips.py:121-223), net='vgg' only.

    lpips=False ("baseline", what --loss_fn vgg_ssl uses with the SimCLR-trained trunk): unit-normalise the channels
    of each of the 5 taps, squared difference, channel sum, spatial mean, summed over taps.  lpips=True: the learned
    non-negative 1x1 `lin` layers weight the channels instead of the plain sum.  normalize -> diff^2 -> lin / sum ->
    spatial mean is one kernel per tap (csrc/lpips.hip); both images go through the trunk as one batch."""

    def __init__(self, pretrained=True, net='vgg', version='0.1', lpips=True, spatial=False, pnet_rand=False,
                 pnet_tune=False, use_dropout=True, model_path='pretrained/lpips_vgg_v0.1.pt', eval_mode=True,
                 verbose=False, pretrained_weights=None):
        super().__init__()
        if net not in ('vgg', 'vgg16'):
            raise NotImplementedError(f'LPIPS trunk {net!r}: GANgealing only uses VGG16 (lpips.py:15,19)')
        if spatial:
            raise NotImplementedError('spatial LPIPS maps are not used on the training path')
        self.pnet_type, self.pnet_tune, self.pnet_rand = net, pnet_tune, pnet_rand
        self.spatial, self.lpips, self.version = spatial, lpips, version
        self.scaling_layer = ScalingLayer()
        self.chns = [64, 128, 256, 512, 512]
        self.L = len(self.chns)
        self.net = vgg16(pretrained=not pnet_rand, requires_grad=pnet_tune, pretrained_weights=pretrained_weights)
        self.lins_loaded = False
        if lpips:
            self.lin0 = NetLinLayer(self.chns[0], use_dropout=use_dropout)
            self.lin1 = NetLinLayer(self.chns[1], use_dropout=use_dropout)
            self.lin2 = NetLinLayer(self.chns[2], use_dropout=use_dropout)
            self.lin3 = NetLinLayer(self.chns[3], use_dropout=use_dropout)
            self.lin4 = NetLinLayer(self.chns[4], use_dropout=use_dropout)
            self.lins = nn.ModuleList([self.lin0, self.lin1, self.lin2, self.lin3, self.lin4])
            if pretrained:
                self.load_state_dict(torch.load(model_path, map_location='cpu'), strict=False)
                self.lins_loaded = True
        if eval_mode:
            self.eval()

    def forward(self, in0, in1, retPerLayer=False, normalize=False):
        if normalize:
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        n = in0.shape[0]
        x = torch.cat([in0, in1], 0)                      # one batched pass for both images
        if self.version == '0.1':
            x = self.scaling_layer(x)
        def lin_of(kk):
            if not self.lpips:
                return None
            if self.training and any(isinstance(m, nn.Dropout) for m in self.lins[kk].model):
                raise NotImplementedError('LPIPS lin layers with active dropout (training mode) are not fused')
            return self.lins[kk].model[-1].weight

        observe = self.__dict__.get('observe')            # diagnostics hook, see vgg16.forward
        if x.dtype == torch.float32 and not ({'lpips_tail', 'lpips_tap'} & conv_mfma.DISABLED):
            res = [v.view(n, 1, 1, 1) for v in self.net(x, tap=lambda kk, f: lpips_tap(f, lin_of(kk)), observe=observe)]
            feats = []
        else:
            feats = self.net(x, observe=observe)
            res = []
        for kk, f in enumerate(feats):
            lin = lin_of(kk)
            if f.dtype == torch.float32 and 'lpips_tail' not in conv_mfma.DISABLED:
                res.append(lpips_tail(f, lin).view(n, 1, 1, 1))
                continue
            f = f / (torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True)) + 1e-10)
            d = (f[:n] - f[n:]) ** 2
            d = d.sum(dim=1, keepdim=True) if lin is None else F.conv2d(d, lin)
            res.append(d.mean(dim=(2, 3), keepdim=True))
        # the reference accumulates IN PLACE into res[0] (lpips.py:203-205: `val = res[0]; val += res[l]`), so with
        # retPerLayer the first "per-layer" entry it hands back is the total - reproduced as is
        val = res[0]
        if retPerLayer:
            for r in res[1:]:
                val += r
            return val, res
        # same left-to-right sum, out of place: an in-place update of a view of a custom node's output makes autograd
        # rebase the view (CopySlices: ~13 clone / copy launches in the backward of every step)
        for r in res[1:]:
            val = val + r
        return val


class _Scaled(nn.Module):
    """loss_fn(x, y) / 18 as a module (get_perceptual_loss wraps vgg_ssl in a lambda, lpips.py:13-17)."""

    def __init__(self, inner, divisor):
        super().__init__()
        self.inner, self.divisor = inner, divisor

    def forward(self, x, y):
        return self.inner(x, y) / self.divisor


VGG_SSL_WEIGHTS = 'pretrained/simclr_vgg_phase150.pt'
LPIPS_WEIGHTS = 'pretrained/lpips_vgg_v0.1.pt'


def get_perceptual_loss(loss_fn, device, weights=None, allow_random=False, trunk_weights=None):
    """lpips.py:11-23.  The reference downloads its weights; here they must already be on disk (`weights`, default the
    reference's `pretrained/...` path).
      vgg_ssl: `weights` = the SimCLR VGG16 `features` state_dict (simclr_vgg_phase150.pt).
      lpips:   `weights` = lpips_vgg_v0.1.pt, which holds ONLY the learned `lin` layers (the reference takes the trunk
               from torchvision's ImageNet VGG16, lpips_backbones.py:101); the trunk comes from `trunk_weights` (a
               torchvision-layout `features` state_dict) unless the file itself carries `net.slice*` entries.
    A missing file - or, for lpips, a trunk that was never loaded - raises unless `allow_random` (synthetic benchmark /
    parity runs: a warning and the seeded random trunk)."""
    import os
    import warnings
    if loss_fn == 'vgg_ssl':
        path = VGG_SSL_WEIGHTS if weights is None else weights
        have = os.path.isfile(path)
        if not have and not allow_random:
            raise FileNotFoundError(f'{path}: SimCLR VGG16 weights not found (allow_random=True / '
                                    f'GANGEALING_SYNTHETIC=1 runs on a seeded random trunk instead)')
        if not have:
            warnings.warn(f'perceptual loss: {path} not found - using a RANDOMLY INITIALISED VGG16 trunk '
                          f'(synthetic benchmark configuration; not a training objective)')
        model = LPIPS(net='vgg', lpips=False, pnet_rand=True, pretrained_weights=path if have else None)
        return _Scaled(model, 18.0).to(device)
    if loss_fn == 'lpips':
        path = LPIPS_WEIGHTS if weights is None else weights
        if not os.path.isfile(path):
            raise FileNotFoundError(f'{path}: LPIPS needs its learned lin layers and an ImageNet VGG16 trunk; neither '
                                    f'can be downloaded here.  Build LPIPS(...) yourself and load a state_dict.')
        sd = torch.load(path, map_location='cpu')
        model = LPIPS(net='vgg', pnet_rand=True, pretrained=False)
        missing_lins = [f'lin{k}.model.1.weight' for k in range(5) if f'lin{k}.model.1.weight' not in sd]
        if missing_lins:
            raise RuntimeError(f'{path}: not an LPIPS checkpoint (missing {missing_lins})')
        model.load_state_dict(sd, strict=False)
        model.lins_loaded = True
        if any(k.startswith('net.slice') for k in sd):
            trunk_keys = {k for k in model.state_dict() if k.startswith('net.slice')}
            absent = sorted(trunk_keys - set(sd))
            if absent:
                raise RuntimeError(f'{path}: incomplete VGG16 trunk (missing {absent[:4]} ...)')
            model.net.weights_loaded = True
        elif trunk_weights is not None:
            model.net.load_features_state_dict(torch.load(trunk_weights, map_location='cpu'), strict=True)
        if not model.net.weights_loaded:
            msg = (f'{path} holds the lin layers only; the VGG16 trunk needs `trunk_weights` (torchvision vgg16 '
                   f'`features` state_dict: the ImageNet weights the reference downloads)')
            if not allow_random:
                raise FileNotFoundError(msg)
            warnings.warn('perceptual loss: ' + msg + ' - using a RANDOMLY INITIALISED trunk (synthetic run)')
        return model.to(device)
    raise NotImplementedError(loss_fn)


# round-1 name (bench / tests): the vgg_ssl form with a random trunk
def VGGPerceptualLoss(seed=0):
    return _Scaled(LPIPS(net='vgg', lpips=False, pnet_rand=True), 18.0)
