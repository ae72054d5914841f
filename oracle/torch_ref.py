"""torch-CPU restatement of the model-level hot path (generator, STN, training loss).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the checker for model-level GPU parity and the
`cpu_baseline` ("port") leg of bench.py.  Never imported by gangealing_amd.

It restates what the reference executes on a CPU-only host - its pure-PyTorch op fallbacks
(upfirdn2d_native upfirdn2d.py:161-200, CPU fused_leaky_relu fused_act.py:87-94, F.conv2d via the
conv2d_gradfix pass-through conv2d_gradfix.py:34-42) composed as networks.py / spatial_transformer.py /
warping_heads.py / antialiased_sampling.py / loss.py compose them - as FUNCTIONS over a state_dict
(keys identical to the reference's), in the reference's own per-sample-weight formulation of the
modulated convolution, so it is an independent check of the product's shared-weight kernels.
Pinned against tests/golden/{generator16,stn,train_step}.npz (reference outputs) by
tests/test_oracle_golden.py.
"""
import math

import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


# ------------------------------------------------------------------ op fallbacks

def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """upfirdn2d_native (upfirdn2d.py:161-200) via zero-stuffing + F.conv2d on a (N*C,1,H,W) view."""
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    u = x.new_zeros(n * c, 1, h * up, w * up)
    u[:, :, ::up, ::up] = x.reshape(n * c, 1, h, w)
    u = F.pad(u, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    u = u[:, :, max(-p0, 0):u.shape[2] - max(-p1, 0), max(-p0, 0):u.shape[3] - max(-p1, 0)]
    out = F.conv2d(u, torch.flip(kernel, [0, 1]).view(1, 1, kh, kw).to(x.dtype))
    out = out[:, :, ::down, ::down]
    return out.reshape(n, c, out.shape[2], out.shape[3])


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=SQRT2):
    shape = [1, -1] + [1] * (x.ndim - 2)
    return F.leaky_relu(x + bias.view(shape), negative_slope) * scale


def blur_kernel(gain=1.0):
    k = torch.tensor([1., 3., 3., 1.])
    k = k[None, :] * k[:, None]
    return k / k.sum() * gain


def equal_linear(sd, prefix, x, lr_mul=1.0, activation=False):
    w = sd[prefix + '.weight']
    scale = (1 / math.sqrt(w.shape[1])) * lr_mul
    b = sd[prefix + '.bias'] * lr_mul
    if activation:
        return fused_leaky_relu(F.linear(x, w * scale), b)
    return F.linear(x, w * scale, b)


# ------------------------------------------------------------------ generator (networks.py:233-586)

def modulated_conv(sd, prefix, x, w_latent, demodulate, upsample):
    weight = sd[prefix + '.weight']                      # (1, Cout, Cin, k, k)
    _, cout, cin, k, _ = weight.shape
    n, _, h, wd = x.shape
    style = equal_linear(sd, prefix + '.modulation', w_latent).view(n, 1, cin, 1, 1)
    wgt = (1 / math.sqrt(cin * k * k)) * weight * style
    if demodulate:
        wgt = wgt * torch.rsqrt(wgt.pow(2).sum([2, 3, 4]) + 1e-8).view(n, cout, 1, 1, 1)
    if upsample:
        wt = wgt.transpose(1, 2).reshape(n * cin, cout, k, k)
        out = F.conv_transpose2d(x.reshape(1, n * cin, h, wd), wt, padding=0, stride=2, groups=n)
        out = out.view(n, cout, out.shape[2], out.shape[3])
        return upfirdn2d(out, blur_kernel(4.0), pad=(1, 1))
    out = F.conv2d(x.reshape(1, n * cin, h, wd), wgt.reshape(n * cout, cin, k, k), padding=k // 2, groups=n)
    return out.view(n, cout, h, wd)


def styled_conv(sd, prefix, x, w_latent, noise, upsample):
    out = modulated_conv(sd, prefix + '.conv', x, w_latent, True, upsample)
    if noise is None:
        noise = torch.randn(out.shape[0], 1, out.shape[2], out.shape[3])
    out = out + sd[prefix + '.noise.weight'] * noise
    return fused_leaky_relu(out, sd[prefix + '.activate.bias'])


def to_rgb(sd, prefix, x, w_latent, skip):
    out = modulated_conv(sd, prefix + '.conv', x, w_latent, False, False) + sd[prefix + '.bias']
    if skip is not None:
        out = out + upfirdn2d(skip, blur_kernel(4.0), up=2, pad=(2, 1))
    return out


def mapping(sd, z, n_mlp=8):
    x = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(n_mlp):
        x = equal_linear(sd, f'style.{i + 1}', x, lr_mul=0.01, activation=True)
    return x


def generator_synthesis(sd, latent, size, noise=None):
    """latent (N, n_latent, D) W+ codes -> image.  noise: list of per-layer tensors or None (random)."""
    log_size = int(math.log2(size))
    num_layers = (log_size - 2) * 2 + 1
    if noise is None:
        noise = [None] * num_layers
    n = latent.shape[0]
    out = sd['input.input'].repeat(n, 1, 1, 1)
    out = styled_conv(sd, 'conv1', out, latent[:, 0], noise[0], False)
    skip = to_rgb(sd, 'to_rgb1', out, latent[:, 1], None)
    i = 1
    for j in range(log_size - 2):
        out = styled_conv(sd, f'convs.{2 * j}', out, latent[:, i], noise[1 + 2 * j], True)
        out = styled_conv(sd, f'convs.{2 * j + 1}', out, latent[:, i + 1], noise[2 + 2 * j], False)
        skip = to_rgb(sd, f'to_rgbs.{j}', out, latent[:, i + 2], skip)
        i += 2
    return skip


def generator(sd, z, size, noise=None):
    w = mapping(sd, z)
    n_latent = 2 * int(math.log2(size)) - 2
    latent = w.unsqueeze(1).repeat(1, n_latent, 1)
    return generator_synthesis(sd, latent, size, noise), latent


# ------------------------------------------------------------------ anti-aliased sampling

def mip_pyramid_stack(x, num_levels):
    """Gaussian stack (N,C,D,H,W) of antialiased_sampling.py:119-149 (power-of-two sizes)."""
    levels = [x]
    cur = x
    c = x.shape[1]
    filt = blur_kernel(1.0)[None, None].repeat(c, 1, 1, 1)
    for i in range(1, num_levels):
        cur = F.conv2d(F.pad(cur, (1, 1, 1, 1), mode='reflect'), filt, stride=2, groups=c)
        levels.append(F.interpolate(cur, scale_factor=2.0 ** i, mode='bilinear', align_corners=False))
    return torch.stack(levels, dim=2)


def mipmap_warp(x, grid, padding_mode, max_num_levels=3.5):
    n, c, h, w = x.shape
    assert float(math.log2(w)).is_integer()
    cx = (w - 1.0) * (grid[..., 0] + 1.0) / 2.0
    cy = (h - 1.0) * (grid[..., 1] + 1.0) / 2.0
    coords = torch.stack([cx, cy], dim=3)
    cp = F.pad(coords.permute(0, 3, 1, 2), (1, 1, 1, 1), mode='replicate').permute(0, 2, 3, 1)
    neigh = [cp[:, 1:-1, :-2], cp[:, 1:-1, 2:], cp[:, :-2, 1:-1], cp[:, 2:, 1:-1]]
    dists = torch.stack([torch.sum((o - coords) ** 2, dim=3).clamp(min=1.0) ** 0.5 for o in neigh])
    levels = torch.log2(dists.max(dim=0).values).clamp(min=0.0, max=max_num_levels - 1.0)
    num_levels = int(levels.max().ceil().item()) + 1
    stack = mip_pyramid_stack(x, num_levels)
    d = stack.shape[2]
    warped = F.grid_sample(stack.reshape(n, c * d, h, w), grid, padding_mode=padding_mode, align_corners=False)
    warped = warped.reshape(n, c, d, warped.shape[2], warped.shape[3])
    lv = levels[:, None, None].expand(n, c, 1, *levels.shape[1:])
    o0 = torch.gather(warped, 2, lv.floor().long())
    o1 = torch.gather(warped, 2, lv.ceil().long())
    return (o0 + (lv % 1.0) * (o1 - o0))[:, :, 0]


def bilinear_downsample(x, stride):
    k = torch.arange(1, 2 * stride + 1, 2, dtype=torch.float64)
    k = torch.cat([k, k.flip(0)])
    k = (k / k.sum()).float()
    c = x.shape[1]
    r = stride // 2
    x = F.pad(x, (r, r, r, r), mode='reflect')
    x = F.conv2d(x, k[None, None, None, :].repeat(c, 1, 1, 1), stride=(1, stride), groups=c)
    return F.conv2d(x, k[None, None, :, None].repeat(c, 1, 1, 1), stride=(stride, 1), groups=c)


# ------------------------------------------------------------------ STN (spatial_transformer.py:388-615)

def conv_layer(sd, prefix, x, downsample=False, activate=True, bias=True):
    """ConvLayer (networks.py:589-635): [Blur] -> EqualConv2d -> [FusedLeakyReLU]; Sequential indices."""
    idx = 0
    if downsample:
        w = sd[f'{prefix}.1.weight']
        k = w.shape[-1]
        p = (4 - 2) + (k - 1)
        x = upfirdn2d(x, blur_kernel(1.0), pad=((p + 1) // 2, p // 2))
        idx = 1
    w = sd[f'{prefix}.{idx}.weight']
    k = w.shape[-1]
    scale = 1 / math.sqrt(w.shape[1] * k * k)
    conv_bias = sd.get(f'{prefix}.{idx}.bias') if (bias and not activate) else None
    x = F.conv2d(x, w * scale, conv_bias, stride=2 if downsample else 1, padding=0 if downsample else k // 2)
    if activate:
        x = fused_leaky_relu(x, sd[f'{prefix}.{idx + 1}.bias'])
    return x


def res_block(sd, prefix, x, downsample):
    out = conv_layer(sd, prefix + '.conv1', x)
    out = conv_layer(sd, prefix + '.conv2', out, downsample=downsample)
    skip = conv_layer(sd, prefix + '.skip', x, downsample=downsample, activate=False, bias=False)
    return (out + skip) / SQRT2


def stn_trunk(sd, prefix, x, flow_size, is_flow):
    log_size = int(math.log2(flow_size))
    end_log = log_size - 4 if is_flow else 2
    out = conv_layer(sd, f'{prefix}convs.0', x)
    n_down = 0
    for bi, _ in enumerate(range(log_size, end_log, -1)):
        down = (not is_flow) or (n_down < 3)
        n_down += down
        out = res_block(sd, f'{prefix}convs.{bi + 1}', out, down)
    return conv_layer(sd, f'{prefix}final_conv', out)


def cluster_classifier(sd, x, size):
    """ResnetClassifier.forward (reference models/cluster_classifier.py:28-48): optional bilinear downsample to
    `size`, 1x1 stem, one down-sampling ResBlock per octave down to 4x4, final 3x3 conv, EqualLinear + fused lrelu."""
    if x.shape[-1] > size:
        x = bilinear_downsample(x, x.shape[-1] // size)
    feats = stn_trunk(sd, '', x, size, False)
    return equal_linear(sd, 'to_logits', feats.reshape(feats.shape[0], -1), activation=True)


def reverse_topk_accuracy(predictions, gt_scores, k=1):
    """models/__init__.py:37-43."""
    top = predictions.argmax(dim=1, keepdim=True)
    return (top == gt_scores.topk(k=k, dim=1).indices).any(dim=1).float().mean()


def similarity_matrix(params):
    rot = torch.tanh(params[:, 0]) * math.pi
    s = torch.exp(params[:, 1])
    c, sn = torch.cos(rot), torch.sin(rot)
    return torch.stack([s * c, -s * sn, params[:, 2], s * sn, s * c, params[:, 3]], dim=1).reshape(-1, 2, 3)


def similarity_stn(sd, prefix, x, source, flow_size, padding_mode, base=None, out_res=None):
    feats = stn_trunk(sd, prefix, x, flow_size, False)
    feats = equal_linear(sd, f'{prefix}final_linear', feats.reshape(feats.shape[0], -1), activation=True)
    params = F.linear(feats, sd[f'{prefix}warp_head.linear.weight'], sd[f'{prefix}warp_head.linear.bias'])
    m = similarity_matrix(params)
    if base is not None:
        bottom = torch.tensor([[[0., 0., 1.]]]).expand(m.shape[0], 1, 3)
        m = base @ torch.cat([m, bottom], 1)
    res = flow_size if out_res is None else out_res
    grid = F.affine_grid(m, (m.shape[0], source.shape[1], res, res), align_corners=False)
    return mipmap_warp(source, grid, padding_mode), grid, m


def flow_stn(sd, prefix, x, source, flow_size, padding_mode, base=None, ds=8):
    feats = stn_trunk(sd, prefix, x, flow_size, True)

    def head(name):
        h = F.conv2d(feats, sd[f'{prefix}warp_head.{name}.0.weight'] / math.sqrt(feats.shape[1] * 9),
                     sd[f'{prefix}warp_head.{name}.0.bias'], padding=1)
        w2 = sd[f'{prefix}warp_head.{name}.2.weight']
        return F.conv2d(F.relu(h), w2 / math.sqrt(w2.shape[1] * 9), sd[f'{prefix}warp_head.{name}.2.bias'], padding=1)

    low, mask = head('flow_out'), head('mask_out')
    n, _, h, w = low.shape
    sm = torch.softmax(mask.view(n, 1, 9, ds, ds, h, w), dim=2)
    nb = F.unfold(ds * low, [3, 3], padding=1).view(n, 2, 9, 1, 1, h, w)
    delta = torch.sum(sm * nb, dim=2).permute(0, 4, 2, 5, 3, 1).reshape(n, ds * h, ds * w, 2)
    eye = torch.eye(2, 3).unsqueeze(0)
    flow = F.affine_grid(eye, (1, 1, ds * h, ds * w), align_corners=False) + delta
    if base is not None:
        ones = torch.ones(n, flow.shape[1] * flow.shape[2], 1)
        flow = (torch.cat([flow.reshape(n, -1, 2), ones], 2) @ base.permute(0, 2, 1)).reshape(flow.shape)
    return mipmap_warp(source, flow, padding_mode), flow, delta


def composed_stn(sd, x, flow_size, supersize, padding_mode, transforms=('similarity', 'flow'), source=None):
    """ComposedSTN.forward (spatial_transformer.py:78-139) for K=1; returns (out, flow_or_matrix)."""
    composed = len(transforms) > 1
    out, warp, src = x, None, (x if source is None else source)
    for i, t in enumerate(transforms):
        prefix = f'stns.{i}.' if composed else ''
        inp = bilinear_downsample(out, out.shape[-1] // flow_size) if out.shape[-1] > flow_size else out
        if t == 'similarity':
            out, grid, warp = similarity_stn(sd, prefix, inp, src, flow_size, padding_mode, base=warp)
        else:
            out, grid, warp = flow_stn(sd, prefix, inp, src, flow_size, padding_mode, base=warp)
    return out, warp


# ------------------------------------------------------------------ losses and the train step

def total_variation_loss(delta):
    def h(a):
        return torch.where(a <= 1.0, 0.5 * a.pow(2), a - 0.5).mean()
    return h((delta[:, :, :-1] - delta[:, :, 1:]).abs()) + h((delta[:, :-1] - delta[:, 1:]).abs())


def latent_interpolate(ll_sd, w, psi, inject, n_latent):
    target = (ll_sd['lat_mean'] + ll_sd['coefficients'] @ ll_sd['directions']).repeat(w.shape[0], 1)
    head = target.lerp(w, psi).unsqueeze(1).repeat(1, inject, 1)
    return torch.cat([head, w.unsqueeze(1).repeat(1, n_latent - inject, 1)], 1)


def train_loss(g_sd, stn_sd, ll_sd, z, gen_size, flow_size, psi, inject, padding_mode, transforms, loss_fn,
               tv_weight, id_weight, noise1=None, noise2=None):
    """gangealing_loss + regularisers (loss.py:64-75, train.py:117-124).  Returns (total, parts)."""
    n_latent = 2 * int(math.log2(gen_size)) - 2
    with torch.no_grad():
        unaligned, latent = generator(g_sd, z, gen_size, noise1)
    w_aligned = latent_interpolate(ll_sd, latent[:, 0], psi, inject, n_latent)
    target = generator_synthesis(g_sd, w_aligned, gen_size, noise2)
    if gen_size > flow_size:
        target = bilinear_downsample(target, gen_size // flow_size)
        stn_in = bilinear_downsample(unaligned, gen_size // flow_size)
    else:
        stn_in = unaligned
    pred, delta = composed_stn(stn_sd, stn_in, flow_size, flow_size, padding_mode, transforms)
    ploss = loss_fn(pred, target).mean()
    total = ploss
    tv = idl = None
    if 'flow' in transforms:
        tv, idl = total_variation_loss(delta), delta.pow(2).mean()
        total = total + tv_weight * tv + id_weight * idl
    return total, dict(p=ploss, tv=tv, f=idl, pred=pred, target=target, unaligned=unaligned, delta=delta)


def mse_loss_fn(a, b):
    return ((a - b) ** 2).mean(dim=(1, 2, 3))
