"""BASELINE.json's own configurations as parity cases.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

One driver, `run_config(api, cfg, device)`, evaluates a training-loss step through an `api` namespace that holds the
reference's public names (Generator, get_stn, DirectionInterpolator, LPIPS, BilinearDownsample, gangealing_loss,
gangealing_cluster_loss, total_variation_loss, flow_identity_loss).  oracle/make_golden_configs.py passes the modules
imported from /root/reference (authoring container, CPU) and stores what comes back under tests/golden/; the GPU tests
pass gangealing_amd's modules and compare - the same code drives both, so the test reads like the reference's
train.py:106-124.  Everything random is replaced by name-keyed deterministic arrays (oracle/det_weights.py): weights,
z (torch.randn is patched for the (batch, dim_latent) draw of loss.py:24) and the per-layer noise images (the
generator is called through NoiseFeeder, which turns `noise=None` into explicit lists).

Configurations:
  c2  LSUN Cats 256^2, similarity+flow STN at 128^2, per-GPU batch 16, vgg_ssl loss form      (the benchmark config)
  c2t the same with textured generator images (noise-injection weights x10)
  c2r the same with a WIDE DYNAMIC RANGE inside the generator: the modulation (style) layers of the 13 styled
      convolutions drawn with scales 1e-5 .. 1e5, so the operand x * style of the modulated convolutions
      (networks.py:243-253) spans 1e-5 .. 3e5 from layer to layer while demodulation keeps the outputs O(1) - what
      trained StyleGAN2 generators do at their high-resolution layers (the reason for the reference's fp16 `normalize`
      branch, networks.py:237-242).  Exercises the exponent range of the split-precision limbs.
  c1  LSUN Cats 64^2, similarity-only STN, batch 4 (BASELINE.json configs[0])
  c4  CelebA-HQ 512^2 flags (scripts/training/celeba.sh:4-6 at gen_size 512): BilinearDownsample(4), border padding,
      inject 6, ndirs 512, sample_from_full_res, tv 2500, LPIPS with lin layers; batch 2
  c5  LSUN Cars clustering (scripts/training/lsun_cars.sh:4-7): num_heads 4, flips, ndirs 5, inject 6,
      sample_from_full_res, reflection padding, tv 2500, LPIPS with lin layers; batch 2 (G pass 2 at 8, flow STN and
      VGG at 16 / 32 images)
"""
import contextlib
import types

import numpy as np
import torch

from .det_weights import det_array, det_state_dict

STN_RULES = (('warp_head.linear', 0.02), ('flow_out.2', 0.02), ('mask_out', 0.5))

CONFIGS = {
    'c2': dict(gen_size=256, flow_size=128, real_size=256, batch=16, transform=['similarity', 'flow'], num_heads=1,
               flips=False, inject=5, ndirs=1, padding_mode='reflection', tv_weight=1000.0, flow_identity_weight=1.0,
               sample_from_full_res=False, loss='vgg_ssl', psi=0.5),
    # c2 with O(1) high-frequency image content: the generator's noise-injection weights drawn with scale 1.0 instead of
    # 0.1 (per-pixel N(0,1) noise enters every layer at full strength: neighbouring pixels of the 256^2 image differ by
    # 0.55 on average, of the 128^2 STN input by 0.17 - a warp error of 1e-3 pixel is then worth ~2e-4)
    'c2t': dict(gen_size=256, flow_size=128, real_size=256, batch=16, transform=['similarity', 'flow'], num_heads=1,
                flips=False, inject=5, ndirs=1, padding_mode='reflection', tv_weight=1000.0, flow_identity_weight=1.0,
                sample_from_full_res=False, loss='vgg_ssl', psi=0.5, gen_rules=(('noise', 1.0),)),
    'c2r': dict(gen_size=256, flow_size=128, real_size=256, batch=16, transform=['similarity', 'flow'], num_heads=1,
                flips=False, inject=5, ndirs=1, padding_mode='reflection', tv_weight=1000.0, flow_identity_weight=1.0,
                sample_from_full_res=False, loss='vgg_ssl', psi=0.5,
                gen_rules=(('conv1.conv.modulation', 1e5), ('convs.0.conv.modulation', 1e-5),
                           ('convs.1.conv.modulation', 1e3), ('convs.2.conv.modulation', 3e-3),
                           ('convs.3.conv.modulation', 3e4), ('convs.4.conv.modulation', 1e-4),
                           ('convs.5.conv.modulation', 30.0), ('convs.6.conv.modulation', 1e4),
                           ('convs.7.conv.modulation', 1e-2), ('convs.8.conv.modulation', 1e2),
                           ('convs.9.conv.modulation', 1e-5), ('convs.10.conv.modulation', 1e5),
                           ('convs.11.conv.modulation', 1e-3))),
    # BASELINE.json configs[0]: 64^2, similarity-only STN, batch 4 (delta_flow is the (N, 2, 3) matrix; no TV /
    # identity terms: train.py only evaluates them for flow STNs)
    'c1': dict(gen_size=64, flow_size=64, real_size=64, batch=4, transform=['similarity'], num_heads=1,
               flips=False, inject=5, ndirs=1, padding_mode='reflection', tv_weight=0.0, flow_identity_weight=0.0,
               sample_from_full_res=False, loss='vgg_ssl', psi=0.5),
    'c4': dict(gen_size=512, flow_size=128, real_size=512, batch=2, transform=['similarity', 'flow'], num_heads=1,
               flips=False, inject=6, ndirs=512, padding_mode='border', tv_weight=2500.0, flow_identity_weight=1.0,
               sample_from_full_res=True, loss='lpips', psi=0.3),
    'c5': dict(gen_size=256, flow_size=128, real_size=256, batch=2, transform=['similarity', 'flow'], num_heads=4,
               flips=True, inject=6, ndirs=5, padding_mode='reflection', tv_weight=2500.0, flow_identity_weight=1.0,
               sample_from_full_res=True, loss='lpips', psi=0.7),
}


# BASELINE configs[3] / [4] at the batch bench.py runs them with (4: `bench.py --workload c4|c5`) and at the per-GPU batch
# of the reference's own recipes on 8 GPUs (scripts/training/celeba.sh:4-6, lsun_cars.sh:4-7: 16): the batch decides
# which tile variant of each convolution kernel is launched, so these are the variants the bench lines time
# (c5 at 16 needs > 62 GB on the CPU in float64: cfg_c5b8 carries float32 + float64, cfg_c5b16 - the batch
# `bench.py --workload c5 --batch 16` and the recipe run - the reference's float32 run alone)
for _base, _batch in (('c4', 4), ('c5', 4), ('c4', 16), ('c5', 8), ('c5', 16)):
    CONFIGS[f'{_base}b{_batch}'] = dict(CONFIGS[_base], batch=_batch)
# C5 at the benched per-GPU batch 16: the reference's run needs > 62 GB on the CPU even in float32 (the STN and the
# perceptual loss see 16 x 4 heads x 2 flips = 128 images), so the reference evaluates the batch in two HALVES of 8
# samples: same name-keyed latents / noise rows as the batch-16 run (`parent` names the tag, `part` the sample range of
# the `full_batch`).  No operation of the step couples samples (no batch statistics; every loss term is a mean over
# samples), so activations of the batch-16 run are the halves' rows and its losses / gradients their averages.
for _k, _part in enumerate(((0, 8), (8, 16))):
    CONFIGS[f'c5b16h{_k}'] = dict(CONFIGS['c5'], batch=8, parent='c5b16', full_batch=16, part=_part)


def T(a, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return t if dtype is None else t.to(dtype)


def det_lpips_state_dict(module):
    """Name-keyed weights for an LPIPS module (reference key names): He-scaled trunk convolutions so the activations
    stay O(1) through 13 layers, small biases, non-negative lin layers (as trained LPIPS lins are)."""
    sd = {}
    for name, p in module.named_parameters():
        if name.startswith('lins.'):
            continue                                   # aliases of lin0..lin4
        if 'lin' in name:
            sd[name] = torch.from_numpy(np.abs(det_array(name, p.shape, 0.2)))
        elif name.endswith('bias'):
            sd[name] = torch.from_numpy(det_array(name, p.shape, 0.05))
        else:
            fan_in = p.shape[1] * p.shape[2] * p.shape[3]
            sd[name] = torch.from_numpy(det_array(name, p.shape, (2.0 / fan_in) ** 0.5))
    return sd


class NoiseFeeder:
    """Generator front that replaces `noise=None` by explicit, name-keyed noise images: call k of the step gets the
    list noises[k] (loss.py:21-29 calls the generator twice per step with fresh noise each time)."""

    def __init__(self, generator, tag, device, dtype=None, part=None, full=None):
        self.generator, self.tag, self.device, self.dtype = generator, tag, device, dtype
        self.n_latent = generator.n_latent
        self.calls = 0
        self.outputs = []
        self.part, self.full = part, full        # samples [lo, hi) of a run of `full` samples (see CONFIGS['c5b16h*'])

    def noise_for(self, call, batch):
        res = lambda i: 2 ** ((i + 5) // 2)
        if self.part is None:
            return [T(det_array(f'{self.tag}.noise{call}.{i}', (batch, 1, res(i), res(i))), self.device, self.dtype)
                    for i in range(self.generator.num_layers)]
        lo, hi = self.part
        m = batch // (hi - lo)                   # rows per sample in this call (sample-major: 1, or num_heads)
        return [T(det_array(f'{self.tag}.noise{call}.{i}', (self.full * m, 1, res(i), res(i)))[lo * m:hi * m],
                  self.device, self.dtype) for i in range(self.generator.num_layers)]

    def __call__(self, styles, noise=None, **kw):
        batch = styles[0].shape[0]
        out = self.generator(styles, noise=self.noise_for(self.calls, batch), **kw)
        self.calls += 1
        self.outputs.append(out[0])
        return out


class Tap:
    """Callable front of a module that records what it returned (the losses only hand back scalars)."""

    def __init__(self, module):
        self.module = module
        self.outputs = []

    def __call__(self, *args, **kw):
        out = self.module(*args, **kw)
        self.outputs.append(out)
        return out


@contextlib.contextmanager
def det_randn(tag, batch, dim_latent, dtype=None, part=None, full=None):
    """torch.randn(batch, dim_latent, ...) -> the name-keyed z of this configuration (loss.py:24); part / full: rows
    [lo, hi) of the z of a `full`-sample run."""
    real = torch.randn

    def fake(*size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        if shape == (batch, dim_latent):
            if part is not None:
                return T(det_array(f'{tag}.z', (full, dim_latent))[part[0]:part[1]], kw.get('device', 'cpu'), dtype)
            return T(det_array(f'{tag}.z', shape), kw.get('device', 'cpu'), dtype)
        return real(*size, **kw)
    torch.randn = fake
    try:
        yield
    finally:
        torch.randn = real


def build_models(api, cfg, tag, device, dtype=None):
    """Generator (frozen), STN, latent learner, perceptual loss and the fake->STN resize, with name-keyed weights.
    dtype=torch.float64 (reference on CPU only) gives the double-precision evaluation the gradient checks use as
    ground truth."""
    gen = api.Generator(cfg['gen_size'], 512, 8, channel_multiplier=2)
    torch.nn.Module.load_state_dict(gen, det_state_dict(gen, cfg.get('gen_rules', ())), strict=False)
    gen = gen.to(device).eval().requires_grad_(False)
    stn = api.get_stn(list(cfg['transform']), flow_size=cfg['flow_size'], supersize=cfg['real_size'],
                      channel_multiplier=0.5, num_heads=cfg['num_heads'])
    torch.nn.Module.load_state_dict(stn, det_state_dict(stn, STN_RULES), strict=False)
    stn = stn.to(device)
    ll = api.DirectionInterpolator(None, cfg['ndirs'], cfg['inject'], gen.n_latent, cfg['num_heads'])
    with torch.no_grad():
        ll.directions.copy_(torch.from_numpy(det_array(f'{tag}.ll.directions', (cfg['ndirs'], 512))))
        ll.lat_mean.copy_(torch.from_numpy(det_array(f'{tag}.ll.lat_mean', (1, 512))))
        ll.coefficients.copy_(torch.from_numpy(det_array(f'{tag}.ll.coefficients', (cfg['num_heads'], cfg['ndirs']),
                                                         0.3)))
    ll = ll.to(device)
    lpips = cfg['loss'] == 'lpips'
    net = api.LPIPS(net='vgg', lpips=lpips, pnet_rand=True, pretrained=False, verbose=False)
    torch.nn.Module.load_state_dict(net, det_lpips_state_dict(net), strict=False)
    net = net.to(device).eval()
    if lpips:
        loss_fn = net
    else:
        loss_fn = lambda x, y: net(x, y) / 18.0                 # lpips.py:17
    factor = cfg['gen_size'] // cfg['flow_size']
    resize = api.BilinearDownsample(factor, 3).to(device) if factor > 1 else torch.nn.Sequential()
    if dtype is not None:
        gen, stn, ll, net, resize = (m.to(dtype) for m in (gen, stn, ll, net, resize))
        for m in stn.modules():                    # the reference keeps identity_flow as a plain tensor attribute
            if isinstance(m.__dict__.get('identity_flow'), torch.Tensor):
                m.identity_flow = m.identity_flow.to(dtype)
    return gen, stn, ll, loss_fn, resize


def run_config(api, name, device, backward=True, dtype=None):
    """One loss evaluation + backward of configuration `name` (train.py:106-124).  -> dict of tensors / floats:
    unaligned, target, pred, delta_flow (the regularised one), ploss, tv, identity, total, grads {param name: grad}."""
    cfg = CONFIGS[name]
    tag = f"cfg.{cfg.get('parent', name)}"
    part, full = cfg.get('part'), cfg.get('full_batch')
    gen, stn, ll, loss_fn, resize = build_models(api, cfg, tag, device, dtype)
    feeder = NoiseFeeder(gen, tag, device, dtype, part, full)
    stn_tap = Tap(stn)
    resize_tap = Tap(resize)
    clustering = cfg['num_heads'] > 1 or cfg['flips']
    with det_randn(tag, cfg['batch'], 512, dtype, part, full):
        common = dict(sample_from_full_res=cfg['sample_from_full_res'], padding_mode=cfg['padding_mode'])
        if clustering:
            ploss, delta = api.gangealing_cluster_loss(feeder, stn_tap, ll, loss_fn, resize_tap, cfg['psi'], cfg['batch'],
                                                       512, False, cfg['num_heads'], cfg['flips'], device, **common)
        else:
            ploss, delta = api.gangealing_loss(feeder, stn_tap, ll, loss_fn, resize_tap, cfg['psi'], cfg['batch'], 512,
                                               False, device, **common)
    if 'flow' in cfg['transform']:
        tv = api.total_variation_loss(delta)
        idl = api.flow_identity_loss(delta)
        total = ploss + cfg['tv_weight'] * tv + cfg['flow_identity_weight'] * idl
    else:                               # train.py:113-123: the flow regularisers exist for flow STNs only
        tv = idl = torch.zeros((), dtype=ploss.dtype, device=ploss.device)
        total = ploss
    out = dict(unaligned=feeder.outputs[0].detach(), target=resize_tap.outputs[0].detach(),
               pred=stn_tap.outputs[0][0].detach(), stn_delta=stn_tap.outputs[0][1].detach(),
               delta_flow=delta.detach(), ploss=ploss.detach(), tv=tv.detach(), identity=idl.detach(),
               total=total.detach())
    if backward:
        params = [(n, p) for n, p in stn.named_parameters()] + [('ll.coefficients', ll.coefficients)]
        grads = torch.autograd.grad(total, [p for _, p in params], allow_unused=True)
        out['grads'] = {n: g.detach() for (n, _), g in zip(params, grads) if g is not None}
    return out


# ---- compact storage of large tensors: the same slices are taken from the reference (when the fixture is written)
# ---- and from the HIP path (when it is compared)

def pack_batch(t, prefix):
    """(N, ...) tensor -> {first sample in full, a strided subsample of every sample, per-sample float64 sums}."""
    t = t.detach().cpu()
    n = t.shape[0]
    flat = t.reshape(n, -1)
    stride = max(1, flat.shape[1] // 16384)
    off = 3 % stride if stride > 1 else 0
    return {f'{prefix}_first': t[0].numpy().copy(), f'{prefix}_sub': flat[:, off::stride].numpy().copy(),
            f'{prefix}_sum': flat.double().sum(1).numpy(), f'{prefix}_abssum': flat.double().abs().sum(1).numpy()}


def pack_grads(grads, prefix='grad_'):
    """{name: grad} -> ({name: norm}, {key: array}) with small gradients in full and a strided sample of large ones."""
    norms, arrays = {}, {}
    for name, g in grads.items():
        g = g.detach().cpu()
        norms[name] = float(g.double().norm())
        flat = g.reshape(-1)
        stride = max(1, flat.numel() // 2048)
        arrays[prefix + name.replace('.', '_')] = flat[::stride].numpy().copy()
    return norms, arrays


def smooth_images(tag, n, size, device='cpu'):
    """(n, 3, size, size) image-like test input: a 32x32 name-keyed random field bilinearly enlarged (smooth structure
    at the scale of generator images) plus 5 % white noise (so that no two neighbouring pixels are equal)."""
    low = torch.from_numpy(det_array(tag + '.low', (n, 3, 32, 32), 0.6))
    x = torch.nn.functional.interpolate(low, size=(size, size), mode='bilinear', align_corners=False)
    return (x + torch.from_numpy(det_array(tag + '.fine', (n, 3, size, size), 0.05))).to(device)


def api_namespace(**kw):
    return types.SimpleNamespace(**kw)


def sample_rows(t, cfg, part):
    """Rows of samples [lo, hi) of a tensor of a `cfg` run whose first dimension is batch (sample), batch x heads
    (sample-major: latent_learner.py:62-63) or flips x batch x heads (loss.py:45-48)."""
    lo, hi = part
    b, h = cfg['batch'], cfg['num_heads']
    n = t.shape[0]
    if n == b:
        return t[lo:hi]
    if n == b * h:
        return t.reshape(b, h, *t.shape[1:])[lo:hi].reshape(-1, *t.shape[1:])
    if n == 2 * b * h:
        return t.reshape(2, b, h, *t.shape[1:])[:, lo:hi].reshape(-1, *t.shape[1:])
    raise ValueError(f'first dimension {n} is not batch / batch x heads / 2 x batch x heads of {b}, {h}')
