"""Generate tests/golden/*.npz by running the REFERENCE's own CPU bodies.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Runs only in the authoring container
(needs /root/reference); the GPU box only ever sees the committed .npz files.

    python oracle/make_golden.py            # rewrites tests/golden/

How the reference is imported: torchvision / lmdb / tensorboard / sklearn-free
stubs are placed in sys.modules, torch.utils.cpp_extension.load is neutralised (the CPU path of
upfirdn2d / fused_leaky_relu never touches the JIT-built CUDA module), and Tensor.cuda is the
identity (FlowHead.__init__ calls .cuda(), warping_heads.py:158).  Nothing is written to
/root/reference.
"""
import json
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('GANGEALING_REFERENCE', '/root/reference')
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle.det_weights import det_array, det_state_dict  # noqa: E402


class _Stub(types.ModuleType):
    """Module stub whose every attribute is a do-nothing callable/class."""

    def __getattr__(self, item):
        if item.startswith('__'):
            raise AttributeError(item)
        return type(item, (), {'__init__': lambda self, *a, **k: None, '__call__': lambda self, *a, **k: None})


def import_reference():
    for name in ['torchvision', 'torchvision.models', 'torchvision.datasets', 'torchvision.datasets.utils',
                 'torchvision.transforms', 'torchvision.utils', 'lmdb', 'tensorboard',
                 'torch.utils.tensorboard', 'moviepy', 'moviepy.editor', 'ray', 'termcolor', 'cv2', 'lpips']:
        if name not in sys.modules:
            m = _Stub(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules['torchvision'].models = sys.modules['torchvision.models']
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules['torchvision'].utils = sys.modules['torchvision.utils']
    sys.modules['torch.utils.tensorboard'].SummaryWriter = object
    sys.modules['torchvision.datasets.utils'].download_url = lambda *a, **k: None
    import torch.utils.cpp_extension as cpp
    cpp.load = lambda *a, **k: types.SimpleNamespace()
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    import models  # noqa: F401  (reference package)
    return models


def save(name, cases):
    flat = {}
    for ci, case in enumerate(cases):
        for k, v in case.items():
            if isinstance(v, (dict, list, tuple, str, int, float, bool)) and not isinstance(v, np.ndarray):
                v = np.frombuffer(json.dumps(v).encode(), dtype=np.uint8)
            elif isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            flat[f'case{ci:02d}/{k}'] = v
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **flat)
    print(f'{name}: {len(cases)} cases, {os.path.getsize(path) / 1024:.1f} KiB')


def rnd(name, shape, scale=1.0, dtype=np.float32):
    return torch.from_numpy(det_array(name, shape, scale, dtype))


# ---------------------------------------------------------------------------------------------

def gen_upfirdn2d():
    from models.stylegan2.op.upfirdn2d import upfirdn2d_native
    from models.stylegan2.networks import make_kernel
    k1331 = make_kernel([1, 3, 3, 1])
    specs = [
        # (N,C,H,W), kernel, up, down, (pad_x0,pad_x1,pad_y0,pad_y1), tag
        ((2, 3, 9, 9), k1331 * 4, 1, 1, (1, 1, 1, 1), 'G blur after up-conv 9->8'),
        ((2, 2, 17, 17), k1331 * 4, 1, 1, (1, 1, 1, 1), 'G blur 17->16'),
        ((1, 2, 33, 33), k1331 * 4, 1, 1, (1, 1, 1, 1), 'G blur 33->32'),
        ((2, 3, 4, 4), k1331 * 4, 2, 1, (2, 1, 2, 1), 'ToRGB skip upsample 4->8'),
        ((1, 3, 16, 16), k1331 * 4, 2, 1, (2, 1, 2, 1), 'ToRGB skip upsample 16->32'),
        ((2, 2, 16, 16), k1331, 1, 1, (2, 2, 2, 2), 'STN blur before 3x3 s2 16->17'),
        ((2, 2, 16, 16), k1331, 1, 1, (1, 1, 1, 1), 'STN blur before 1x1 s2 16->15'),
        ((1, 2, 8, 8), k1331, 1, 2, (1, 1, 1, 1), 'Downsample (bwd of upsample) 8->4'),
        ((1, 2, 67, 70), k1331, 1, 1, (2, 2, 2, 2), 'multi-tile non-square'),
        ((1, 1, 5, 7), torch.from_numpy(det_array('k34', (3, 4))), 1, 1, (2, 1, 1, 2), 'asymmetric 3x4 taps'),
        ((1, 2, 6, 6), k1331, 2, 2, (2, 1, 2, 1), 'up=down=2'),
        ((1, 1, 9, 9), k1331, 1, 1, (-1, 0, 0, -2), 'negative pad (crop)'),
        ((2, 1, 1, 1), k1331 * 4, 2, 1, (2, 1, 2, 1), '1x1 input'),
        ((1, 1, 6, 6), torch.from_numpy(det_array('k55', (5, 5))), 3, 2, (3, 2, 3, 2), 'generic 5x5 up3 down2'),
        ((1, 1, 8, 8), torch.from_numpy(det_array('k22', (2, 2))), 1, 2, (0, 0, 0, 0), '2x2 down2 (mode 6)'),
    ]
    cases = []
    for i, (shape, k, up, down, pad, tag) in enumerate(specs):
        x = rnd(f'upfirdn.x{i}', shape).requires_grad_(True)
        out = upfirdn2d_native(x, k, up, up, down, down, *pad)
        g = rnd(f'upfirdn.g{i}', out.shape)
        (gx,) = torch.autograd.grad(out, x, g)
        cases.append(dict(x=x, k=k, out=out, g=g, gx=gx,
                          meta=dict(up=up, down=down, pad=list(pad), tag=tag)))
    save('upfirdn2d', cases)


def gen_fused_act():
    from models.stylegan2.op.fused_act import fused_leaky_relu
    cases = []
    for i, shape in enumerate([(4, 8), (2, 6, 5, 5), (3, 16, 8, 8), (1, 4, 3, 7), (2, 5, 1, 1)]):
        x = rnd(f'fused.x{i}', shape).requires_grad_(True)
        b = rnd(f'fused.b{i}', (shape[1],), 0.5).requires_grad_(True)
        if i == 1:  # plant exact zeros of x+b and of x
            with torch.no_grad():
                x[0, 0, 0, 0] = -b[0]
                x[0, 1, 0, 0] = 0.0
        out = fused_leaky_relu(x, b)          # CPU branch, fused_act.py:87-94 (slope hard-coded 0.2)
        g = rnd(f'fused.g{i}', shape)
        gx, gb = torch.autograd.grad(out, (x, b), g)
        cases.append(dict(x=x, b=b, out=out, g=g, gx=gx, gb=gb, meta=dict(negative_slope=0.2, scale=2 ** 0.5)))
    save('fused_act', cases)


def make_grids(n, r):
    """A family of sampling grids (N,r,r,2) exercising identity / zoom / rotation / smooth flow."""
    def aff(theta):
        return F.affine_grid(torch.tensor(theta, dtype=torch.float32).view(1, 2, 3).repeat(n, 1, 1),
                             (n, 3, r, r), align_corners=False)
    c, s = math.cos(0.6), math.sin(0.6)
    smooth = aff([[1, 0, 0], [0, 1, 0]])
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, r), torch.linspace(-1, 1, r), indexing='ij')
    smooth = smooth + 0.35 * torch.stack([torch.sin(3 * yy + 1) * xx, torch.cos(2 * xx) * yy], -1)[None]
    return {
        'identity': aff([[1, 0, 0], [0, 1, 0]]),
        'zoom_in_0.5': aff([[0.5, 0, 0.1], [0, 0.5, -0.2]]),
        'zoom_out_2': aff([[2, 0, 0], [0, 2, 0]]),
        'zoom_out_4': aff([[4, 0, 0.3], [0, 4, 0]]),
        'zoom_out_6_rot': aff([[6 * c, -6 * s, 0], [6 * s, 6 * c, 0]]),
        'rot_shift': aff([[1.3 * c, -1.3 * s, 0.4], [1.3 * s, 1.3 * c, -0.3]]),
        'smooth_flow': smooth,
        'jitter': aff([[1, 0, 0], [0, 1, 0]]) + rnd('jitter', (n, r, r, 2), 0.01),
    }


def gen_mipmap_warp():
    from models.spatial_transformers.antialiased_sampling import MipmapWarp
    cases = []
    ci = 0
    for (s, r, modes, names) in [
        (32, 16, ['border', 'reflection', 'zeros'], None),
        (64, 32, ['reflection'], ['zoom_out_4', 'smooth_flow', 'jitter']),
        (30, 16, ['border'], ['zoom_out_2', 'rot_shift']),          # non power of two -> pad path
        (16, 16, ['reflection'], ['identity', 'zoom_out_6_rot']),
    ]:
        grids = make_grids(2, r)
        for name, grid in grids.items():
            if names is not None and name not in names:
                continue
            for mode in modes:
                x = rnd(f'mip.x{ci}', (2, 3, s, s)).requires_grad_(True)
                grid_l = grid.clone().requires_grad_(True)
                warp = MipmapWarp(max_num_levels=3.5)
                out = warp(x, grid_l, padding_mode=mode)
                g = rnd(f'mip.g{ci}', out.shape)
                gx, ggrid = torch.autograd.grad(out, (x, grid_l), g)
                cases.append(dict(x=x, grid=grid_l, out=out, g=g, gx=gx, ggrid=ggrid,
                                  levels_map=warp.levels_map,
                                  meta=dict(padding_mode=mode, grid=name, max_num_levels=3.5)))
                ci += 1
    save('mipmap_warp', cases)
    # plain (non anti-aliased) Warp == F.grid_sample, all padding modes
    cases = []
    for ci, mode in enumerate(['border', 'reflection', 'zeros']):
        grids = make_grids(1, 12)
        for name in ['zoom_out_2', 'rot_shift', 'smooth_flow']:
            x = rnd(f'gs.x{ci}{name}', (1, 2, 10, 14))
            out = F.grid_sample(x, grids[name], padding_mode=mode, align_corners=False)
            cases.append(dict(x=x, grid=grids[name], out=out, meta=dict(padding_mode=mode, grid=name)))
    save('grid_sample', cases)


def gen_mipmap_warp_deep():
    """MipmapWarp beyond the heads' 3.5: the reference's DEFAULT constructor (max_num_levels = 8,
    antialiased_sampling.py:22) and 4.5, on grids that reach the deep levels (the 7-level clamp included), a non
    power-of-two image (pad path) and a 32-pixel image whose pyramid ends at 1 x 1 before the clamp does."""
    from models.spatial_transformers.antialiased_sampling import MipmapWarp

    def aff(theta, n, r):
        return F.affine_grid(torch.tensor(theta, dtype=torch.float32).view(1, 2, 3).repeat(n, 1, 1),
                             (n, 3, r, r), align_corners=False)
    c, s = math.cos(0.4), math.sin(0.4)
    cases = []
    ci = 0
    for (size, r, mnl, mode, name, theta) in [
        (128, 16, 8, 'border', 'zoom_out_4', [[4, 0, 0.3], [0, 4, 0]]),
        (128, 16, 8, 'reflection', 'zoom_out_12_rot', [[12 * c, -12 * s, 0], [12 * s, 12 * c, 0.2]]),
        (128, 8, 8, 'reflection', 'zoom_out_40', [[40, 0, 0], [0, 40, 0]]),          # levels hit the clamp at 7
        (128, 8, 8, 'zeros', 'zoom_out_20_rot', [[20 * c, -20 * s, 0.1], [20 * s, 20 * c, 0]]),
        (128, 16, 8, 'border', 'aniso', [[9, 0, 0], [0, 1.5, 0]]),
        (128, 16, 8, 'border', 'default_ctor_identity', [[1, 0, 0], [0, 1, 0]]),
        (100, 16, 8, 'reflection', 'pad_zoom_out_8', [[8, 0, 0], [0, 8, 0.1]]),       # 100 -> reflect-padded to 128
        (32, 16, 8, 'border', 'small_zoom_out_4', [[4, 0, 0], [0, 4, 0]]),            # 32-pixel base: 6 levels exist
        (64, 32, 4.5, 'reflection', 'zoom_out_6_rot', [[6 * c, -6 * s, 0], [6 * s, 6 * c, 0]]),
        (64, 16, 4.5, 'border', 'zoom_out_3', [[3, 0, 0.2], [0, 3, -0.1]]),
        (64, 16, 4.5, 'zeros', 'zoom_out_16', [[16, 0, 0], [0, 16, 0]]),              # clamp at 3.5
    ]:
        grid = aff(theta, 2, r)
        if name != 'default_ctor_identity':
            grid = grid + rnd(f'mipd.j{ci}', (2, r, r, 2), 0.02)
        x = rnd(f'mipd.x{ci}', (2, 3, size, size)).requires_grad_(True)
        grid_l = grid.clone().requires_grad_(True)
        warp = MipmapWarp() if name == 'default_ctor_identity' else MipmapWarp(max_num_levels=mnl)
        out = warp(x, grid_l, padding_mode=mode)
        g = rnd(f'mipd.g{ci}', out.shape)
        gx, ggrid = torch.autograd.grad(out, (x, grid_l), g)
        cases.append(dict(x=x, grid=grid_l, out=out, g=g, gx=gx, ggrid=ggrid, levels_map=warp.levels_map,
                          meta=dict(padding_mode=mode, grid=name, max_num_levels=mnl,
                                    max_level_reached=float(warp.levels_map.detach().max() * (mnl - 1.0)))))
        ci += 1
    save('mipmap_warp_deep', cases)


def gen_heads():
    from models.spatial_transformers.warping_heads import SimilarityHead, FlowHead, apply_affine
    cases = []
    # similarity: params -> matrix -> composed with base -> affine_grid
    head = SimilarityHead(8)
    params = rnd('sim.params', (3, 4), 0.5).requires_grad_(True)
    matrix = head.make_affine_matrix(*torch.split(params, 1, dim=1))          # (N,1,2,3)
    base = rnd('sim.base', (3, 1, 2, 3), 0.7)
    composed = base @ head.make_3x3(matrix)
    grid = F.affine_grid(composed.reshape(3, 2, 3), (3, 3, 16, 16), align_corners=False)
    g = rnd('sim.g', grid.shape)
    (gparams,) = torch.autograd.grad(grid, params, g)
    cases.append(dict(params=params, matrix=matrix.reshape(3, 2, 3), base=base.reshape(3, 2, 3),
                      composed=composed.reshape(3, 2, 3), grid=grid, g=g, gparams=gparams, meta=dict(kind='similarity')))
    save('similarity_head', cases)

    cases = []
    for ci, (n, h) in enumerate([(2, 4), (1, 16)]):
        fh = FlowHead((1, 8, h, h))
        low = rnd(f'flow.low{ci}', (n, h, h, 2), 0.05).requires_grad_(True)
        mask = rnd(f'flow.mask{ci}', (n, 9 * 64, h, h), 1.0).requires_grad_(True)
        base = rnd(f'flow.base{ci}', (n, 2, 3), 0.6).requires_grad_(True)
        delta = fh.upsample_flow(low, mask)
        flow = fh.identity_flow + delta
        flow = apply_affine(base, flow)
        g1 = rnd(f'flow.g1{ci}', flow.shape)
        g2 = rnd(f'flow.g2{ci}', delta.shape)
        glow, gmask, gbase = torch.autograd.grad([flow, delta], (low, mask, base), [g1, g2])
        res = flow.size(1)
        resized = F.interpolate(flow.permute(0, 3, 1, 2), scale_factor=2 * res / flow.size(2),
                                mode='bilinear').permute(0, 2, 3, 1)
        cases.append(dict(low=low, mask=mask, base=base, delta=delta, flow=flow, identity=fh.identity_flow,
                          g_flow=g1, g_delta=g2, glow=glow, gmask=gmask, gbase=gbase, resized2x=resized,
                          meta=dict(kind='flow', ds=8)))
    save('flow_head', cases)


def gen_misc():
    from models.spatial_transformers.antialiased_sampling import BilinearDownsample
    from models.losses.loss import total_variation_loss, flow_identity_loss
    cases = []
    for ci, (stride, s) in enumerate([(2, 16), (4, 32), (2, 64)]):
        x = rnd(f'bd.x{ci}', (2, 3, s, s)).requires_grad_(True)
        out = BilinearDownsample(stride, 3)(x)
        g = rnd(f'bd.g{ci}', out.shape)
        (gx,) = torch.autograd.grad(out, x, g)
        cases.append(dict(x=x, out=out, g=g, gx=gx, meta=dict(stride=stride)))
    save('bilinear_downsample', cases)
    cases = []
    for ci, scale in enumerate([0.3, 3.0]):
        d = rnd(f'tv.d{ci}', (2, 16, 16, 2), scale).requires_grad_(True)
        tv = total_variation_loss(d)
        idl = flow_identity_loss(d)
        (gd,) = torch.autograd.grad(1000.0 * tv + idl, d)
        cases.append(dict(delta=d, tv=tv, identity=idl, gdelta=gd, meta=dict(tv_weight=1000.0, id_weight=1.0)))
    save('flow_losses', cases)


def gen_modconv():
    from models.stylegan2.networks import ModulatedConv2d, StyledConv, ToRGB
    cases = []
    for ci, (cin, cout, k, up, demod, n, h) in enumerate([
        (8, 6, 3, False, True, 2, 8),
        (8, 6, 3, True, True, 2, 4),
        (6, 3, 1, False, False, 2, 8),
        (16, 16, 3, False, True, 3, 5),
    ]):
        m = ModulatedConv2d(cin, cout, k, 12, demodulate=demod, upsample=up)
        m.load_state_dict(det_state_dict(m), strict=False)
        x = rnd(f'mc.x{ci}', (n, cin, h, h)).requires_grad_(True)
        w = rnd(f'mc.w{ci}', (n, 12)).requires_grad_(True)
        out = m(x, w)
        g = rnd(f'mc.g{ci}', out.shape)
        gx, gw = torch.autograd.grad(out, (x, w), g)
        style = m.modulation(w)
        cases.append(dict(x=x, w=w, out=out, g=g, gx=gx, gw=gw, style=style,
                          weight=m.weight[0], mod_weight=m.modulation.weight, mod_bias=m.modulation.bias,
                          meta=dict(cin=cin, cout=cout, k=k, upsample=up, demodulate=demod)))
    save('modulated_conv', cases)


def gen_generator():
    from models.stylegan2.networks import Generator
    g = Generator(16, 512, 8)
    g.load_state_dict(det_state_dict(g), strict=False)
    g.eval()
    z = rnd('gen.z', (2, 512))
    noise = [rnd(f'gen.noise{i}', (2, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2))) for i in range(g.num_layers)]
    img, latent = g([z], return_latents=True, noise=noise)
    w = latent[:, 0].detach().clone().requires_grad_(True)
    latent2 = w.unsqueeze(1).repeat(1, g.n_latent, 1)
    img2, _ = g([latent2], input_is_latent=True, noise=noise)
    gimg = rnd('gen.gimg', img2.shape)
    (gw,) = torch.autograd.grad(img2, w, gimg)
    save('generator16', [dict(z=z, img=img, w=latent[:, 0], gimg=gimg, gw=gw, img_from_w=img2,
                              **{f'noise{i}': t for i, t in enumerate(noise)},
                              meta=dict(size=16, style_dim=512, n_mlp=8, num_layers=g.num_layers))])


def gen_stn():
    from models.spatial_transformers.spatial_transformer import get_stn
    from models.losses.loss import total_variation_loss, flow_identity_loss
    rules = (('warp_head.linear', 0.02), ('flow_out.2', 0.02), ('mask_out', 0.5))
    cases = []
    for ci, (transforms, flow_size, supersize, mode) in enumerate([
        (['similarity'], 32, 32, 'border'),
        (['similarity', 'flow'], 64, 128, 'reflection'),
    ]):
        stn = get_stn(transforms, flow_size=flow_size, supersize=supersize, channel_multiplier=0.5, num_heads=1)
        sd = det_state_dict(stn, rules)
        torch.nn.Module.load_state_dict(stn, sd, strict=False)
        x = rnd(f'stn.x{ci}', (2, 3, supersize, supersize), 0.5)
        kwargs = dict(return_flow=True, padding_mode=mode)
        if supersize > flow_size:
            kwargs['input_img_for_sampling'] = x
        out, flow_or_m = stn(x, **kwargs)
        loss = (out ** 2).mean()
        if 'flow' in transforms:
            loss = loss + 10.0 * total_variation_loss(flow_or_m) + flow_identity_loss(flow_or_m)
        names = [n for n, _ in stn.named_parameters()]
        grads = torch.autograd.grad(loss, list(stn.parameters()), allow_unused=True)
        gnorm = {n: (float(g.double().norm()) if g is not None else None) for n, g in zip(names, grads)}
        keep = {n.replace('.', '_'): g for n, g in zip(names, grads)
                if g is not None and g.numel() <= 4096 and ('warp_head' in n or 'bias' in n)}
        cases.append(dict(x=x, out=out, flow_or_matrix=flow_or_m, loss=loss,
                          meta=dict(transforms=transforms, flow_size=flow_size, supersize=supersize,
                                    padding_mode=mode, scale_rules=[list(r) for r in rules], grad_norms=gnorm),
                          **{'grad_' + k: v for k, v in keep.items()}))
    save('stn', cases)


def gen_cluster_classifier():
    from models.cluster_classifier import ResnetClassifier
    from models import accuracy
    cases = []
    for ci, (size, supersize, heads) in enumerate([(32, None, 4), (32, 64, 6)]):
        net = ResnetClassifier(size, channel_multiplier=0.5, num_heads=heads, supersize=supersize)
        sd = det_state_dict(net, (('to_logits', 0.05),))
        torch.nn.Module.load_state_dict(net, sd, strict=False)
        res = supersize or size
        x = rnd(f'cls.x{ci}', (5, 3, res, res), 0.5)
        logits = net(x)
        labels = torch.tensor([1, 3, 0, 2, 1]) % heads
        loss = torch.nn.functional.cross_entropy(logits, labels)
        names = [n for n, _ in net.named_parameters()]
        grads = torch.autograd.grad(loss, list(net.parameters()))
        gnorm = {n: float(g.double().norm()) for n, g in zip(names, grads)}
        scores = rnd(f'cls.scores{ci}', (5, heads))
        flipped, _, classes, flip_ixs = net.run_flip(x)
        kept, kept_logits = net.run(x, 1)
        cart, policy = net.run_flip_cartesian(x[:2])
        cases.append(dict(x=x, logits=logits, labels=labels, loss=loss, scores=scores,
                          acc1=accuracy(logits, scores), acc2=accuracy(logits, scores, k=2),
                          grad_to_logits_weight=grads[names.index('to_logits.weight')],
                          grad_to_logits_bias=grads[names.index('to_logits.bias')],
                          assign=net.assign(x), assign_noflip=net.assign(x, ignore_flips=True),
                          run_flip_images=flipped, run_flip_classes=classes, run_flip_ixs=flip_ixs,
                          run_kept=kept, run_kept_logits=kept_logits, cart_images=cart, cart_policy=policy,
                          meta=dict(size=size, supersize=supersize, num_heads=heads, channel_multiplier=0.5,
                                    scale_rules=[['to_logits', 0.05]], grad_norms=gnorm)))
    save('cluster_classifier', cases)


def gen_point_transfer():
    """Key-point helpers of the STN (spatial_transformer.py:141-295, 617-720) on the reference modules."""
    from models.spatial_transformers.spatial_transformer import get_stn
    rules = (('warp_head.linear', 0.05), ('flow_out.2', 0.05), ('mask_out', 0.5))
    cases = []
    for ci, (transforms, flow_size, supersize) in enumerate([(['similarity'], 32, 32), (['similarity', 'flow'], 64, 64)]):
        stn = get_stn(transforms, flow_size=flow_size, supersize=supersize, channel_multiplier=0.5, num_heads=1)
        sd = det_state_dict(stn, rules)
        torch.nn.Module.load_state_dict(stn, sd, strict=False)
        stn.eval()
        imgA = rnd(f'pt.a{ci}', (2, 3, supersize, supersize), 0.5)
        imgB = rnd(f'pt.b{ci}', (2, 3, supersize, supersize), 0.5)
        pts = (rnd(f'pt.p{ci}', (2, 7, 2), 1.0).abs() * 0.3 * supersize + 0.2 * supersize).clamp(0, supersize - 1)
        pts_n = rnd(f'pt.pn{ci}', (2, 7, 2), 0.4).clamp(-0.9, 0.9)
        with torch.no_grad():
            kw = dict(padding_mode='border')
            unc = stn.uncongeal_points(imgB, pts_n, **kw)
            con = stn.congeal_points(imgA, pts, **kw)
            tra = stn.transfer_points(imgA, imgB, pts, **kw)
            case = dict(imgA=imgA, imgB=imgB, points=pts, points_norm=pts_n, uncongealed=unc,
                        congealed=con.float(), transferred=tra,
                        meta=dict(transforms=transforms, flow_size=flow_size, supersize=supersize,
                                  scale_rules=[list(r) for r in rules]))
            if 'flow' in transforms:
                out, warp, flow, inputs, flips = stn.forward_with_flip(imgA, return_flow=True, return_warp=True,
                                                                        return_inputs=True, return_flip_indices=True, **kw)
                a2, b2, pa2, pb2, pick = stn.match_flows(imgA, imgB, pts, pts.flip(1), **kw)
                case.update(fwf_out=out, fwf_warp=warp, fwf_flow=flow, fwf_flips=flips, mf_pointsA=pa2, mf_pointsB=pb2,
                            mf_pick=pick)
        cases.append(case)
    save('point_transfer', cases)


def gen_train_step():
    """One gangealing_loss evaluation (loss.py:64-75) at a plumbing-sized config with an MSE
    stand-in for the VGG loss (torchvision is not installed).  RNG-dependent inputs (z, per-layer
    noise) are made deterministic by patching randn / normal_ draws through explicit arguments."""
    from models.stylegan2.networks import Generator
    from models.spatial_transformers.spatial_transformer import get_stn
    from models.latent_learner import DirectionInterpolator
    from models.spatial_transformers.antialiased_sampling import BilinearDownsample
    from models.losses.loss import total_variation_loss, flow_identity_loss
    gen = Generator(64, 512, 8)
    gen.load_state_dict(det_state_dict(gen), strict=False)
    gen.eval()
    rules = (('warp_head.linear', 0.02), ('flow_out.2', 0.02), ('mask_out', 0.5))
    stn = get_stn(['similarity', 'flow'], flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=1)
    torch.nn.Module.load_state_dict(stn, det_state_dict(stn, rules), strict=False)
    ll = DirectionInterpolator(None, 2, 3, gen.n_latent)
    ll.directions.copy_(rnd('ll.directions', (2, 512)))
    ll.lat_mean.copy_(rnd('ll.lat_mean', (1, 512)))
    with torch.no_grad():
        ll.coefficients.copy_(rnd('ll.coefficients', (1, 2), 0.3))
    resize = torch.nn.Sequential()   # gen_size == flow_size (train.py:62)
    psi = 0.5
    z = rnd('ts.z', (2, 512))
    noise1 = [rnd(f'ts.n1.{i}', (2, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2))) for i in range(gen.num_layers)]
    noise2 = [rnd(f'ts.n2.{i}', (2, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2))) for i in range(gen.num_layers)]
    # sample_gan_supervised_pairs, loss.py:21-29, with explicit noise
    unaligned, w = gen([z], noise=noise1, return_latents=True)
    w_aligned = ll([w[:, 0, :]], psi=psi)
    target, _ = gen(w_aligned, input_is_latent=True, noise=noise2)
    target = resize(target)
    pred, delta = stn(resize(unaligned), return_flow=True, input_img_for_sampling=None, padding_mode='reflection')
    ploss = ((pred - target) ** 2).mean(dim=(1, 2, 3)).mean()
    tv = total_variation_loss(delta)
    idl = flow_identity_loss(delta)
    total = ploss + 1000.0 * tv + 1.0 * idl
    params = list(stn.parameters()) + [ll.coefficients]
    grads = torch.autograd.grad(total, params)
    names = [n for n, _ in stn.named_parameters()] + ['ll.coefficients']
    gnorm = {n: float(g.double().norm()) for n, g in zip(names, grads)}
    save('train_step', [dict(z=z, unaligned=unaligned, target=target, pred=pred, delta=delta,
                             ploss=ploss, tv=tv, identity=idl, total=total, g_coefficients=grads[-1],
                             meta=dict(gen_size=64, flow_size=64, psi=psi, inject=3, ndirs=2, tv_weight=1000.0,
                                       flow_identity_weight=1.0, padding_mode='reflection',
                                       scale_rules=[list(r) for r in rules], grad_norms=gnorm))])


# splat2d: tests/golden/splat2d.npz is written by oracle/make_golden_splat.py ON A GPU BOX from the reference's own
# kernel (oracle/_ref/libsplat_ref.so = utils/splat2d_cuda/src/splat_gpu_impl.cu compiled unmodified by oracle/Makefile);
# this script never touches it.


def gen_stn_inference():
    """Inference options of the STN (spatial_transformer.py:471-567, warping_heads.py:129-131,257-260,280-310):
    iterated similarity warps with intermediates, output_resolution, out-of-bounds detection with and without
    image_bounds, for a single STN and the composed one."""
    from models.spatial_transformers.spatial_transformer import get_stn
    rules = (('warp_head.linear', 0.004), ('flow_out.2', 0.05), ('mask_out', 0.5))
    cases = []
    for ci, transforms in enumerate((['similarity'], ['similarity', 'flow'])):
        stn = get_stn(transforms, flow_size=64, supersize=128, channel_multiplier=0.5, num_heads=1)
        torch.nn.Module.load_state_dict(stn, det_state_dict(stn, rules), strict=False)
        stn.eval()
        x = rnd(f'inf.x{ci}', (3, 3, 128, 128), 0.5)
        bounds = torch.tensor([[96.0, 128.0], [128.0, 80.0], [128.0, 128.0]])
        with torch.no_grad():
            if len(transforms) == 1:
                out3, grid3, m3, oob3 = stn(x, iters=3, return_warp=True, return_flow=True, return_out_of_bounds=True,
                                            output_resolution=96, padding_mode='border')
                out1, oob1 = stn(x, return_out_of_bounds=True, padding_mode='reflection')
                # image_bounds: the reference compares (N, H*W) with an (N,) threshold vector
                # (warping_heads.py:306-307), which only broadcasts for one image at a time - as the pre-processing
                # application calls it
                oob_b = torch.cat([stn(x[i:i + 1], return_out_of_bounds=True, padding_mode='border',
                                       image_bounds=bounds[i:i + 1])[1] for i in range(3)])
                outs, mats = stn(x, iters=3, return_intermediates=True, padding_mode='border')
                case = dict(x=x, bounds=bounds, out3=out3, grid3=grid3, m3=m3, oob3=oob3, out1=out1, oob1=oob1,
                            oob_b=oob_b, inter_out=torch.stack(outs), inter_m=torch.stack(mats))
            else:       # the composed STN iterates its similarity stage; it has no out-of-bounds output (:117)
                out3, grid3, d3 = stn(x, iters=3, return_warp=True, return_flow=True, output_resolution=96,
                                      padding_mode='border')
                imgs, warps = stn(x, iters=2, return_intermediates=True, padding_mode='reflection')
                case = dict(x=x, out3=out3, grid3=grid3, m3=d3, inter_out=torch.stack(imgs), inter_m=torch.stack(warps))
            case['meta'] = dict(transforms=transforms, scale_rules=[list(r) for r in rules])
        cases.append(case)
    save('stn_inference', cases)


def gen_applications():
    """The per-batch body of the mixed-reality loop (applications/mixed_reality.py:147-218) run with the REFERENCE's
    own functions on CPU: applications.determine_flips + STN.uncongeal_points + un-mirroring + crop offsets, and the
    congealed frames, for (0) a composed STN without classifier on non-square frames ('unimodal': flips inferred by
    forward_with_flip), (1) a K = 2 clustering STN with its classifier, one frame at a time ('predict_cluster').  The
    splat overlay itself needs the GPU kernel and is pinned elsewhere (tests/golden/splat2d.npz, point_transfer.npz).
    ('fixed_cluster' creates its tensors with device='cuda' in the reference and cannot run on the CPU.)"""
    import types as _types
    import applications as ref_app
    from models.spatial_transformers.spatial_transformer import get_stn, SpatialTransformer
    from models.cluster_classifier import ResnetClassifier
    from prepare_data import nchw_center_crop
    rules = (('warp_head.linear', 0.01), ('flow_out.2', 0.05), ('mask_out', 0.5))
    cases = []
    for ci, heads in enumerate((1, 2)):
        args = _types.SimpleNamespace(transform=['similarity', 'flow'], flow_size=64, stn_channel_multiplier=0.5,
                                      num_heads=heads, real_size=128, iters=1 if heads > 1 else 2,
                                      padding_mode='border', no_flip_inference=False)
        t = get_stn(args.transform, flow_size=64, supersize=128, channel_multiplier=0.5, num_heads=heads)
        torch.nn.Module.load_state_dict(t, det_state_dict(t, rules), strict=False)
        t.eval()
        classifier = None
        if heads > 1:
            classifier = ResnetClassifier(64, channel_multiplier=0.5, num_heads=2 * heads, supersize=128)
            torch.nn.Module.load_state_dict(classifier, det_state_dict(classifier, (('to_logits', 0.05),)), strict=False)
            classifier.eval()
        nframes = 3
        frames = rnd(f'app.frames{ci}', (nframes, 3, 128, 160 if heads == 1 else 128), 0.5)
        pts_px = [torch.from_numpy(np.abs(det_array(f'app.points{ci}.{k}', (1, 7, 2), 36.0)) + 12.0).clamp(0, 127)
                  for k in range(heads)]
        pts_norm = [SpatialTransformer.normalize(p, 128, 128) for p in pts_px]
        out_pts, out_flip, out_cluster, out_cong = [], [], [], []
        with torch.no_grad():
            batches = [frames] if heads == 1 else [frames[i:i + 1] for i in range(nframes)]
            for batch in batches:
                n = batch.size(0)
                original = batch
                y0 = x0 = 0
                if batch.size(2) != batch.size(3):
                    batch, (y0, x0) = nchw_center_crop(batch)
                flipped, flip_idx, policy, active = ref_app.determine_flips(args, t, classifier, batch, cluster=None,
                                                                            return_cluster_assignments=True)
                pin = pts_norm[active.item()] if heads > 1 else pts_norm[0].repeat(n, 1, 1)
                prop = t.uncongeal_points(flipped, pin, normalize_input_points=False, warp_policy=policy,
                                          padding_mode=args.padding_mode, iters=args.iters)
                prop[:, :, 0] = torch.where(flip_idx.view(-1, 1), args.real_size - 1 - prop[:, :, 0], prop[:, :, 0])
                prop[:, :, 0] += x0
                prop[:, :, 1] += y0
                if heads > 1:
                    flipped, policy = classifier.run_flip_cartesian(batch)
                cong = t(flipped, output_resolution=args.real_size, warp_policy=policy, unfold=heads > 1,
                         padding_mode=args.padding_mode, iters=args.iters)
                out_pts.append(prop)
                out_flip.append(flip_idx.reshape(-1))
                out_cluster.append(active.reshape(-1))
                out_cong.append(cong if heads > 1 else cong.unsqueeze(1))
        cases.append(dict(frames=frames, points_norm=torch.cat(pts_norm, 0), points=torch.cat(out_pts, 0),
                          flip=torch.cat(out_flip, 0), clusters=torch.cat(out_cluster, 0), congealed=torch.cat(out_cong, 0),
                          meta=dict(num_heads=heads, args={k: v for k, v in vars(args).items()},
                                    stn_rules=[list(r) for r in rules])))
    save('applications', cases)


# ---------------------------------------------------------------------------------------------
# integer by-products of the anti-aliased sampling ("bit-exact warp grid indices")

def aten_source_index(g, size, mode):
    """grid_sampler_compute_source_index (ATen/native/GridSampler.h:27-35,58-60,89-105,143-160), align_corners=False,
    evaluated with float32 torch ops in ATen's order."""
    size_f = torch.tensor(float(size), dtype=torch.float32)
    x = ((g + 1) * size_f - 1) / 2
    if mode == 'border':
        x = torch.minimum(torch.tensor(float(size - 1)), torch.maximum(x, torch.tensor(0.0)))
    elif mode == 'reflection':
        twice_low, twice_high = -1, 2 * size - 1
        mn = torch.tensor(twice_low / 2.0, dtype=torch.float32)
        span = torch.tensor((twice_high - twice_low) / 2.0, dtype=torch.float32)
        x = (x - mn).abs()
        extra = torch.fmod(x, span)
        flips = torch.floor(x / span).to(torch.int64)
        x = torch.where(flips % 2 == 0, extra + mn, span - extra + mn)
        x = torch.minimum(torch.tensor(float(size - 1)), torch.maximum(x, torch.tensor(0.0)))
    return x


def _check_indices_against_aten(grid, size, mode, ix, iy):
    """Authoring-time self check of aten_source_index against the real F.grid_sample: (i) the bilinear sample built
    from floor(ix), floor(iy) reproduces grid_sample's output; (ii) so does the derivative with respect to the grid
    on a piecewise-LINEAR-free random image - at integer coordinates the one-sided derivative depends on which cell
    floor() selected, so a wrong floor shows up as an O(1) error there."""
    n, r = grid.shape[0], grid.shape[1]
    img = rnd(f'idx.img{size}', (n, 2, size, size))
    gd = grid.clone().requires_grad_(True)
    out = F.grid_sample(img, gd, padding_mode=mode, align_corners=False)          # float32: ATen's own index arithmetic
    gout = rnd(f'idx.gout{size}', tuple(out.shape))
    (gg,) = torch.autograd.grad(out, gd, gout)
    x0, y0 = torch.floor(ix).long(), torch.floor(iy).long()
    tx, ty = (ix - x0).double(), (iy - y0).double()
    img = img.double()

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < size) & (yy >= 0) & (yy < size)
        v = img[torch.arange(n)[:, None, None], :, yy.clamp(0, size - 1), xx.clamp(0, size - 1)]   # (n,r,r,c)
        return v * ok[..., None]
    nw, ne, sw, se = tap(y0, x0), tap(y0, x0 + 1), tap(y0 + 1, x0), tap(y0 + 1, x0 + 1)
    manual = (nw * ((1 - tx) * (1 - ty))[..., None] + ne * (tx * (1 - ty))[..., None] +
              sw * ((1 - tx) * ty)[..., None] + se * (tx * ty)[..., None]).permute(0, 3, 1, 2)
    assert float((manual - out.detach().double()).abs().max()) < 1e-4
    if mode == 'zeros':       # no clip / reflect multipliers: d ix / d gx = size / 2 everywhere
        dix = ((ne - nw) * (1 - ty)[..., None] + (se - sw) * ty[..., None])
        man_gx = (dix * gout.double().permute(0, 2, 3, 1)).sum(-1) * (size / 2.0)
        err = (man_gx - gg[..., 0].double()).abs()
        assert float(err.max()) < 1e-3, float(err.max())


def special_grid(size, r, seed):
    """(1, r, r, 2) grid whose source coordinates hit integers, half-integers, the reflection seams (-0.5, size-0.5,
    and their images further out), each also nudged by +-1 ulp of the grid value; the rest is uniform noise over
    [-2.5, 2.5] (far outside the image on both sides)."""
    halves = np.arange(-2 * size, 3 * size + 0.5, 0.5, dtype=np.float64)
    g = ((2 * halves + 1) / size - 1).astype(np.float32)        # exact for power-of-two sizes
    vals = np.concatenate([g, np.nextafter(g, np.float32(np.inf)), np.nextafter(g, np.float32(-np.inf))])
    rs = np.random.RandomState(seed)
    total = r * r
    fill = (rs.rand(max(total - len(vals), 0)) * 5 - 2.5).astype(np.float32)
    gx = np.concatenate([vals, fill])[:total]
    gy = gx.copy()
    rs.shuffle(gx)
    rs.shuffle(gy)
    return torch.from_numpy(np.stack([gx, gy], -1).reshape(1, r, r, 2))


def level_edge_grids(size):
    """(B, 2, 2, 2) grids for size = 2^k + 1 (level coordinates are then 2^(k-1) * (g + 1): an exact scaling).  In
    sample b the two columns are the points (0, 0) and (dx, dy) (both rows equal), so every pixel's neighbour
    distance is sqrt(dx^2 + dy^2): dx is the base distance (1, 2, 4: level exactly 0 / 1 / 2) or one float below it,
    and dy is tuned in float32 so that dx^2 + dy^2 lands on each of the floats next to base^2 (-6 .. +6 ulps) - the
    points where floor / ceil of the level can flip.  Plus plain distances across the whole range and the 2.5 clamp."""
    assert (size - 1) & (size - 2) == 0
    half = np.float32((size - 1) / 2.0)
    f32 = np.float32

    def coord(gv):                       # antialiased_sampling.py:192-193 in float32
        return f32(f32(f32(size - 1) * f32(gv + f32(1))) / f32(2))

    def step(v, k):
        for _ in range(abs(k)):
            v = np.nextafter(v, f32(np.inf) if k > 0 else f32(-np.inf))
        return v
    points, hit = [], []
    for base in (1.0, 2.0, 4.0):
        for k in range(-6, 7):
            target_sq = step(f32(base * base), k)
            gx = f32(base / half - 1.0)
            if k < 0:
                gx = step(gx, -1)
            dx = coord(gx) - coord(f32(-1.0))
            need = float(target_sq) - float(f32(dx * dx))
            if need < 0:
                continue
            guess = f32(np.sqrt(need) / half - 1.0)
            for st in range(0, 200):
                for sgn in (1, -1):
                    gy = step(guess, sgn * st)
                    dy = coord(gy) - coord(f32(-1.0))
                    if f32(f32(dx * dx) + f32(dy * dy)) == target_sq:
                        points.append((gx, gy))
                        hit.append((base, k))
                        break
                else:
                    continue
                break
    for d in (0.25, 0.999, 1.0, 1.5, 2.0, 3.0, 4.0, 5.0, 5.6568542, 5.66, 6.0, 8.0, 11.0):
        points.append((f32(d / half - 1.0), f32(-1.0)))
    g = np.full((len(points), 2, 2, 2), -1.0, dtype=np.float32)
    for b, (gx, gy) in enumerate(points):
        g[b, :, 1] = (gx, gy)
    return torch.from_numpy(g), hit


def gen_warp_indices():
    from models.spatial_transformers.antialiased_sampling import MipmapWarp
    cases = []
    specs = []
    for size, r in ((32, 32), (64, 48)):
        specs.append((f'special{size}', special_grid(size, r, size), size))
    for name, grid in make_grids(2, 24).items():
        specs.append((name, grid, 32))
    specs.append(('zoom_out_4_nonpow2', make_grids(1, 16)['zoom_out_4'], 30))
    for size in (17, 33):
        grids, hit = level_edge_grids(size)
        assert len(hit) >= 30, hit               # the ulp-neighbourhoods of level 0 / 1 / 2 were actually reached
        specs.append((f'level_edges{size}', grids, size))
    max_num_levels, min_level = 3.5, 0.0
    for name, grid, size in specs:
        grid = grid.float().contiguous()
        coords = MipmapWarp._get_coordinates(grid, size, size)
        levels = MipmapWarp._get_mipmap_levels(coords, max_num_levels).clamp(min=min_level)
        num_levels = int(levels.max().ceil().item()) + 1                      # antialiased_sampling.py:52
        for mode in ('border', 'reflection', 'zeros'):
            ix = aten_source_index(grid[..., 0], size, mode)
            iy = aten_source_index(grid[..., 1], size, mode)
            _check_indices_against_aten(grid, size, mode, ix, iy)
            cases.append(dict(grid=grid, ix_nw=torch.floor(ix).to(torch.int32), iy_nw=torch.floor(iy).to(torch.int32),
                              level_floor=levels.floor().to(torch.int32), level_ceil=levels.ceil().to(torch.int32),
                              levels=levels,
                              meta=dict(name=name, size=size, padding_mode=mode, max_num_levels=max_num_levels,
                                        min_level=min_level, num_levels=num_levels)))
    save('warp_indices', cases)


# ---------------------------------------------------------------------------------------------

def gen_annealing():
    """psi annealing, lr_cycle_iters and the DecayingCosineAnnealingWarmRestarts learning-rate sequence exactly as
    train.py drives them (:92-97,129-132), on a short mock schedule."""
    from utils.annealing import DecayingCosineAnnealingWarmRestarts, lr_cycle_iters, get_psi_annealing_fn
    import contextlib
    import io
    cases = []
    for (anneal_psi, period, iters, tm, decay, base_lr) in [(10, 7.5, 120, 2, 0.9, 1e-3), (4, 3.0, 40, 1, 0.8, 1e-2),
                                                            (6, 5.0, 200, 3, 0.5, 1.0)]:
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=base_lr)
        sched = DecayingCosineAnnealingWarmRestarts(opt, T_0=1, T_mult=tm, decay=decay)
        lrs, psis_cos, psis_lin = [], [], []
        cos, lin = get_psi_annealing_fn('cosine'), get_psi_annealing_fn('linear')
        for i in range(1, iters + 1):
            lrs.append(opt.param_groups[0]['lr'])             # the rate the optimizer uses at iteration i
            if i <= anneal_psi:
                psis_cos.append(cos(i, 1.0, 0.0, anneal_psi).item())
                psis_lin.append(lin(i, 1.0, 0.0, anneal_psi).item())
            else:
                sched.step(max(0, (i - anneal_psi) / period))
        zero_iters = []
        if tm > 1:                                            # (the reference divides by log(1) for tm == 1)
            with contextlib.redirect_stdout(io.StringIO()):
                zero_iters = lr_cycle_iters(anneal_psi, period, iters, tm)
        cases.append(dict(lrs=np.array(lrs, dtype=np.float64), psi_cosine=np.array(psis_cos, dtype=np.float64),
                          psi_linear=np.array(psis_lin, dtype=np.float64), zero_lr_iters=np.array(zero_iters),
                          meta=dict(anneal_psi=anneal_psi, period=period, iters=iters, tm=tm, decay=decay,
                                    base_lr=base_lr, sched_state=sched.state_dict()['T_i'])))
    save('annealing', cases)


def gen_half_ops():
    """binary16 inputs through the reference's CPU bodies (the CUDA kernels dispatch half: upfirdn2d_kernel.cu:311,
    fused_bias_act_kernel.cu:89; the CPU bodies are the runnable specification).  Stored as float16."""
    from models.stylegan2.op.upfirdn2d import upfirdn2d_native
    from models.stylegan2.op.fused_act import fused_leaky_relu
    from models.stylegan2.networks import make_kernel
    k1331 = make_kernel([1, 3, 3, 1])
    cases = []
    specs = [((2, 3, 9, 9), k1331 * 4, 1, 1, (1, 1, 1, 1)), ((2, 3, 8, 8), k1331 * 4, 2, 1, (2, 1, 2, 1)),
             ((2, 2, 16, 16), k1331, 1, 1, (2, 2, 2, 2)), ((1, 2, 8, 8), k1331, 1, 2, (1, 1, 1, 1)),
             ((1, 2, 37, 41), k1331, 1, 1, (2, 2, 2, 2)), ((1, 1, 6, 6), k1331, 2, 2, (2, 1, 2, 1))]
    for i, (shape, k, up, down, pad) in enumerate(specs):
        x = rnd(f'half.ux{i}', shape).half().requires_grad_(True)
        out = upfirdn2d_native(x, k.half(), up, up, down, down, *pad)
        g = rnd(f'half.ug{i}', out.shape).half()
        (gx,) = torch.autograd.grad(out, x, g)
        # fp32 evaluation of the same (rounded) inputs: the yardstick for "how far is half arithmetic from exact"
        out32 = upfirdn2d_native(x.detach().float(), k.half().float(), up, up, down, down, *pad)
        cases.append(dict(kind='upfirdn2d', x=x, k=k, out=out, out32=out32, g=g, gx=gx,
                          meta=dict(kind='upfirdn2d', up=up, down=down, pad=list(pad))))
    for i, shape in enumerate([(4, 8), (2, 6, 5, 5), (3, 16, 8, 8), (2, 4, 3, 7)]):
        x = rnd(f'half.fx{i}', shape).half().requires_grad_(True)
        b = rnd(f'half.fb{i}', (shape[1],), 0.5).half().requires_grad_(True)
        out = fused_leaky_relu(x, b)
        g = rnd(f'half.fg{i}', shape).half()
        gx, gb = torch.autograd.grad(out, (x, b), g)
        cases.append(dict(x=x, b=b, out=out, g=g, gx=gx, gb=gb,
                          meta=dict(kind='fused_leaky_relu', negative_slope=0.2, scale=2 ** 0.5)))
    save('half_ops', cases)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import_reference()
    only = sys.argv[1:]
    gens = dict(upfirdn2d=gen_upfirdn2d, fused_act=gen_fused_act, mipmap_warp=gen_mipmap_warp,
                mipmap_warp_deep=gen_mipmap_warp_deep, heads=gen_heads,
                misc=gen_misc, modconv=gen_modconv, generator=gen_generator, stn=gen_stn,
                train_step=gen_train_step, cluster_classifier=gen_cluster_classifier,
                point_transfer=gen_point_transfer, warp_indices=gen_warp_indices, annealing=gen_annealing, stn_inference=gen_stn_inference,
                half_ops=gen_half_ops, applications=gen_applications)
    for name, fn in gens.items():
        if only and name not in only:
            continue
        with torch.enable_grad():
            fn()
