"""Golden vectors at BASELINE.json's own configurations, produced by the REFERENCE on CPU.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py); authoring container only (needs /root/reference).

    python oracle/make_golden_configs.py [c2_generator c2_stn c1 c2 c2t c2r c4 c5 c4b4 c5b4 c4b16 c5b8 c5b16 lpips lpips_masks act_masks stn_decisions]      # ~15 min on 8 cores for all

Writes tests/golden/{c2_generator,c2_stn,cfg_c1,cfg_c2,cfg_c2t,cfg_c4,cfg_c5,lpips}.npz.  The reference runs unmodified: its
modules are imported exactly as oracle/make_golden.py does, plus a local VGG16 `features` stack placed where
lpips_backbones.py:101 asks torchvision for one (torchvision is not installed; the layer list is torchvision's
cfg 'D').  The loss steps are driven by oracle/config_cases.run_config - the same function the GPU tests call with
gangealing_amd's modules.
"""
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle.make_golden import import_reference, save, rnd          # noqa: E402
from oracle.det_weights import det_state_dict                       # noqa: E402
from oracle import config_cases as cc                               # noqa: E402


def local_vgg16(pretrained=False, **kwargs):
    """Stand-in for torchvision.models.vgg16: only `.features` is used (lpips_backbones.py:101)."""
    cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
    layers, cin = [], 3
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return types.SimpleNamespace(features=nn.Sequential(*layers))


def reference_api():
    import_reference()
    sys.modules['torchvision.models'].vgg16 = local_vgg16
    from models.stylegan2.networks import Generator
    from models.spatial_transformers.spatial_transformer import get_stn
    from models.spatial_transformers.antialiased_sampling import BilinearDownsample
    from models.latent_learner import DirectionInterpolator
    from models.losses.lpips import LPIPS
    from models.losses.loss import (gangealing_loss, gangealing_cluster_loss, total_variation_loss,
                                    flow_identity_loss)
    return cc.api_namespace(Generator=Generator, get_stn=get_stn, BilinearDownsample=BilinearDownsample,
                            DirectionInterpolator=DirectionInterpolator, LPIPS=LPIPS, gangealing_loss=gangealing_loss,
                            gangealing_cluster_loss=gangealing_cluster_loss, total_variation_loss=total_variation_loss,
                            flow_identity_loss=flow_identity_loss)


def gen_config(api, name, fp64=True):
    t0 = time.time()
    res = cc.run_config(api, name, 'cpu')
    case = {}
    for key in ('unaligned', 'target', 'pred', 'stn_delta', 'delta_flow'):
        case.update(cc.pack_batch(res[key], key))
    norms, arrays = cc.pack_grads(res['grads'])
    case.update(arrays)
    for key in ('ploss', 'tv', 'identity', 'total'):
        case[key] = res[key]
    # the same step in float64: ground truth for the gradient checks (how far is the reference's own float32
    # evaluation from it?  the HIP path is held to a multiple of that distance).  fp64=False (c5b16: the float64 run of
    # C5 at batch 16 needs more than the authoring container's 62 GB): the fixture holds the float32 run only and the
    # test compares gradients with it directly (tests/test_gpu_configs.py::check_grads_fp32_only)
    norms64 = None
    if fp64:
        res64 = cc.run_config(api, name, 'cpu', dtype=torch.float64)
        norms64, arrays64 = cc.pack_grads(res64['grads'], 'grad64_')
        case.update(arrays64)
        case['total64'] = res64['total']
    case['meta'] = dict(config=name, cfg=cc.CONFIGS[name], grad_norms=norms, grad_norms64=norms64,
                        seconds=round(time.time() - t0, 1),
                        shapes={k: list(res[k].shape) for k in ('unaligned', 'target', 'pred', 'stn_delta', 'delta_flow')})
    save(f'cfg_{name}', [case])
    print(f'  {name}: {time.time() - t0:.0f} s, total loss {float(res["total"]):.6f}')


def gen_config_halves(api, name):
    """cfg_<name>.npz with one case per half of the batch (oracle/config_cases.py: CONFIGS['c5b16h*']): the reference
    cannot hold the whole batch on this host; the test assembles the batch-16 expectations from the halves."""
    import numpy as np
    cases = []
    for k in (0, 1):
        gen_config(api, f'{name}h{k}')
        path = os.path.join(REPO, 'tests', 'golden', f'cfg_{name}h{k}.npz')
        z = np.load(path)
        cases.append({key.split('/', 1)[1]: z[key] for key in z.files})
        os.remove(path)
    import json
    for c in cases:
        c['meta'] = json.loads(bytes(c['meta'].tolist()).decode())
    save(f'cfg_{name}', cases)


def gen_c2_generator(api, n=16):
    """Generator(256) forward + the gradient with respect to w through every style path (networks.py:514-586)."""
    g = api.Generator(256, 512, 8, channel_multiplier=2)
    torch.nn.Module.load_state_dict(g, det_state_dict(g), strict=False)
    g.eval().requires_grad_(False)
    z = rnd('c2gen.z', (n, 512))
    noise = [rnd(f'c2gen.noise{i}', (n, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2))) for i in range(g.num_layers)]
    with torch.no_grad():
        img, latent = g([z], return_latents=True, noise=noise)
    w = latent[:, 0].detach().clone().requires_grad_(True)
    img2, _ = g([w.unsqueeze(1).repeat(1, g.n_latent, 1)], input_is_latent=True, noise=noise)
    gimg = rnd('c2gen.gimg', img2.shape)
    (gw,) = torch.autograd.grad(img2, w, gimg)
    g64 = g.double()
    w64 = latent[:, 0].detach().double().clone().requires_grad_(True)
    img64, _ = g64([w64.unsqueeze(1).repeat(1, g.n_latent, 1)], input_is_latent=True, noise=[t.double() for t in noise])
    (gw64,) = torch.autograd.grad(img64, w64, gimg.double())
    case = dict(z=z, w=latent[:, 0], gw=gw, gw64=gw64, meta=dict(size=256, batch=n, num_layers=g.num_layers))
    case.update(cc.pack_batch(img, 'img'))
    case.update(cc.pack_batch(img2, 'img_from_w'))
    save('c2_generator', [case])


def gen_c2_stn(api, n=16):
    """similarity+flow STN at the benchmark shapes (regression at 128^2, 256^2 input): forward + all gradients."""
    from models.losses.loss import total_variation_loss, flow_identity_loss
    cases = []
    for ci, (full_res, mode) in enumerate([(False, 'reflection'), (True, 'border')]):
        case = {}
        meta = dict(batch=n, padding_mode=mode, sample_from_full_res=full_res)
        for dt, prefix in ((torch.float32, 'grad_'), (torch.float64, 'grad64_')):
            stn = api.get_stn(['similarity', 'flow'], flow_size=128, supersize=256, channel_multiplier=0.5, num_heads=1)
            torch.nn.Module.load_state_dict(stn, det_state_dict(stn, cc.STN_RULES), strict=False)
            stn = stn.to(dt)
            for m in stn.modules():
                if isinstance(m.__dict__.get('identity_flow'), torch.Tensor):
                    m.identity_flow = m.identity_flow.to(dt)
            x = cc.smooth_images(f'c2stn.x{ci}', n, 256).to(dt)
            small = api.BilinearDownsample(2, 3).to(dt)(x)
            # train.py feeds the 128^2 resized fake; with --sample_from_full_res the 256^2 image is the sampling source
            out, flow = stn(small, return_flow=True, padding_mode=mode, input_img_for_sampling=x if full_res else None)
            gout = rnd(f'c2stn.g{ci}', out.shape).to(dt)
            loss = (out * gout).mean() + 10.0 * total_variation_loss(flow) + flow_identity_loss(flow)
            params = list(stn.named_parameters())
            grads = torch.autograd.grad(loss, [p for _, p in params])
            norms, arrays = cc.pack_grads({n_: g for (n_, _), g in zip(params, grads)}, prefix)
            case.update(arrays)
            if dt == torch.float32:
                meta['grad_norms'] = norms
                case['loss'] = loss
                case.update(cc.pack_batch(out, 'out'))
                case.update(cc.pack_batch(flow, 'flow'))
            else:
                meta['grad_norms64'] = norms
                case['loss64'] = loss
        case['meta'] = meta
        cases.append(case)
    save('c2_stn', cases)


def gen_lpips(api):
    """The reference LPIPS class (lpips.py:121-223) in both forms: baseline sum (vgg_ssl) and learned lin layers."""
    cases = []
    for lp in (False, True):
        net = api.LPIPS(net='vgg', lpips=lp, pnet_rand=True, pretrained=False, verbose=False)
        torch.nn.Module.load_state_dict(net, cc.det_lpips_state_dict(net), strict=False)
        net.eval()
        in0 = rnd('lpips.in0', (3, 3, 64, 64), 0.5).requires_grad_(True)
        in1 = rnd('lpips.in1', (3, 3, 64, 64), 0.5)
        val, per_layer = net(in0, in1, retPerLayer=True)
        g = rnd('lpips.g', val.shape)
        (gin0,) = torch.autograd.grad(val, in0, g)
        net64 = net.double()
        in0_64 = in0.detach().double().requires_grad_(True)
        val64 = net64(in0_64, in1.double())
        (gin0_64,) = torch.autograd.grad(val64, in0_64, g.double())
        net.float()
        cases.append(dict(in0=in0, in1=in1, val=val, g=g, gin0=gin0, gin0_64=gin0_64, val64=val64,
                          per_layer=torch.cat([p.reshape(3, 1) for p in per_layer], 1),
                          meta=dict(lpips=lp, state_dict_keys=sorted(net.state_dict().keys()),
                                    shapes={k: list(v.shape) for k, v in net.state_dict().items()})))
    save('lpips', cases)


def _record_lrelu_signs(root):
    """Forward hooks on every FusedLeakyReLU module below `root`: -> (list filled in call order with (out > 0), handles)."""
    from models.stylegan2.op import FusedLeakyReLU
    signs, hooks = [], []
    for mod in root.modules():
        if isinstance(mod, FusedLeakyReLU):
            hooks.append(mod.register_forward_hook(lambda m, i, o: signs.append((o > 0).clone())))
    return signs, hooks


def gen_act_masks(api):
    """The reference's own leaky-ReLU branch decisions (one bit per activation unit, in call order) for two small runs, with
    the gradients of the same runs in float32 and float64: tests/test_gpu_act_masks.py replays the HIP path with these
    decisions pinned (the generator / STN counterpart of lpips_masks).
      case 0  Generator(64), batch 2: gradient of <image, g> w.r.t. w through all ten style inputs (the run of
              gen_c2_generator at a size whose masks fit a fixture);
      case 1  similarity + flow STN at 64^2, batch 4: warped output, flow, every parameter gradient (the run of gen_c2_stn)."""
    from models.losses.loss import total_variation_loss, flow_identity_loss
    cases = []
    # ---- generator
    n = 2
    g = api.Generator(64, 512, 8, channel_multiplier=2)
    torch.nn.Module.load_state_dict(g, det_state_dict(g), strict=False)
    g.eval().requires_grad_(False)
    z = rnd('actmask.gen.z', (n, 512))
    noise = [rnd(f'actmask.gen.noise{i}', (n, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2))) for i in range(g.num_layers)]
    with torch.no_grad():
        _, latent = g([z], return_latents=True, noise=noise)
    w = latent[:, 0].detach().clone().requires_grad_(True)
    signs, hooks = _record_lrelu_signs(g)
    img, _ = g([w.unsqueeze(1).repeat(1, g.n_latent, 1)], input_is_latent=True, noise=noise)
    for h in hooks:
        h.remove()
    gimg = rnd('actmask.gen.gimg', img.shape)
    (gw,) = torch.autograd.grad(img, w, gimg)
    g64 = g.double()
    w64 = latent[:, 0].detach().double().clone().requires_grad_(True)
    img64, _ = g64([w64.unsqueeze(1).repeat(1, g.n_latent, 1)], input_is_latent=True, noise=[t.double() for t in noise])
    (gw64,) = torch.autograd.grad(img64, w64, gimg.double())
    case = dict(w=latent[:, 0], gw=gw, gw64=gw64, img=img.detach(),
                meta=dict(kind='generator', size=64, batch=n, num_layers=g.num_layers,
                          shapes=[list(sg.shape) for sg in signs],
                          active=[float(sg.float().mean()) for sg in signs]))
    for k, sg in enumerate(signs):
        case[f'sign{k:02d}'] = np.packbits(sg.numpy().reshape(-1))
    cases.append(case)
    # ---- STN
    n = 4
    case, meta = {}, dict(kind='stn', batch=n, padding_mode='reflection')
    for dt, prefix in ((torch.float32, 'grad_'), (torch.float64, 'grad64_')):
        stn = api.get_stn(['similarity', 'flow'], flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=1)
        torch.nn.Module.load_state_dict(stn, det_state_dict(stn, cc.STN_RULES), strict=False)
        stn = stn.to(dt)
        for m in stn.modules():
            if isinstance(m.__dict__.get('identity_flow'), torch.Tensor):
                m.identity_flow = m.identity_flow.to(dt)
        x = cc.smooth_images('actmask.stn.x', n, 64).to(dt)
        signs, hooks = _record_lrelu_signs(stn) if dt == torch.float32 else ([], [])
        out, flow = stn(x, return_flow=True, padding_mode='reflection')
        for h in hooks:
            h.remove()
        gout = rnd('actmask.stn.g', out.shape).to(dt)
        loss = (out * gout).mean() + 10.0 * total_variation_loss(flow) + flow_identity_loss(flow)
        params = list(stn.named_parameters())
        grads = torch.autograd.grad(loss, [p for _, p in params])
        norms, arrays = cc.pack_grads({n_: g_ for (n_, _), g_ in zip(params, grads)}, prefix)
        case.update(arrays)
        if dt == torch.float32:
            meta['grad_norms'] = norms
            meta['shapes'] = [list(sg.shape) for sg in signs]
            meta['active'] = [float(sg.float().mean()) for sg in signs]
            case['loss'] = loss
            case.update(cc.pack_batch(out, 'out'))
            case.update(cc.pack_batch(flow, 'flow'))
            for k, sg in enumerate(signs):
                case[f'sign{k:02d}'] = np.packbits(sg.numpy().reshape(-1))
        else:
            meta['grad_norms64'] = norms
    case['meta'] = meta
    cases.append(case)
    save('act_masks', cases)


def gen_stn_decisions(api):
    """The decisions of the STN run of act_masks (case 1: similarity + flow STN at 64^2, batch 4, same weights, same
    input, same loss - asserted) that act_masks does NOT hold, so that tests/test_gpu_stn_decisions.py can pin ALL of them:
      * head_sign{k}   the (N, C) output sign of the similarity stage's final EqualLinear(activation='fused_lrelu')
                       (spatial_transformer.py:458,596 -> networks.py:147-149: a function call, no module to hook);
      * relu{k}        the output sign of the RAFT heads' plain ReLUs (warping_heads.py:160-169), in call order;
      * level_arg{k}   for each anti-aliased warp (MipmapWarp.get_max_coord_distance, antialiased_sampling.py:62-97) the
                       index torch.max(dists, dim=0) returned - the neighbour (0 l, 1 r, 2 u, 3 d) the level's
                       sub-gradient flows through.  Under a similarity warp the four distances are exactly tied in real
                       arithmetic: this index is last-ulp noise of the grid, in the reference too."""
    from models.losses.loss import total_variation_loss, flow_identity_loss
    import models.stylegan2.networks as ref_networks
    n = 4
    stn = api.get_stn(['similarity', 'flow'], flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=1)
    torch.nn.Module.load_state_dict(stn, det_state_dict(stn, cc.STN_RULES), strict=False)
    x = cc.smooth_images('actmask.stn.x', n, 64)
    head_signs, relu_signs, level_args = [], [], []
    real_flr, real_max = ref_networks.fused_leaky_relu, torch.max

    def flr(inp, bias, *a, **k):
        out = real_flr(inp, bias, *a, **k)
        if out.dim() == 2:
            head_signs.append((out > 0).clone())
        return out

    def tmax(*a, **k):
        res = real_max(*a, **k)
        if len(a) == 1 and k.get('dim') == 0 and a[0].dim() == 4 and a[0].shape[0] == 4:
            level_args.append(res[1].clone())
        return res
    hooks = [m.register_forward_hook(lambda mod, i, o: relu_signs.append((o > 0).clone()))
             for m in stn.modules() if isinstance(m, nn.ReLU)]
    ref_networks.fused_leaky_relu, torch.max = flr, tmax
    try:
        out, flow = stn(x, return_flow=True, padding_mode='reflection')
    finally:
        ref_networks.fused_leaky_relu, torch.max = real_flr, real_max
        for h in hooks:
            h.remove()
    gout = rnd('actmask.stn.g', out.shape)
    loss = (out * gout).mean() + 10.0 * total_variation_loss(flow) + flow_identity_loss(flow)
    ref = np.load(os.path.join(REPO, 'tests', 'golden', 'act_masks.npz'))
    assert np.array_equal(loss.detach().numpy(), ref['case01/loss']), 'not the run stored in act_masks.npz'
    assert len(head_signs) == 1 and len(level_args) == 2 and len(relu_signs) == 2, \
        (len(head_signs), len(level_args), len(relu_signs))
    case = dict(meta=dict(batch=n, head_shapes=[list(t.shape) for t in head_signs],
                          relu_shapes=[list(t.shape) for t in relu_signs],
                          level_arg_shapes=[list(t.shape) for t in level_args],
                          level_arg_histogram=[np.bincount(t.numpy().reshape(-1), minlength=4).tolist() for t in level_args]))
    for k, t in enumerate(head_signs):
        case[f'head_sign{k}'] = np.packbits(t.numpy().reshape(-1))
    for k, t in enumerate(relu_signs):
        case[f'relu{k}'] = np.packbits(t.numpy().reshape(-1))
    for k, t in enumerate(level_args):
        case[f'level_arg{k}'] = t.numpy().astype(np.uint8)
    save('stn_decisions', [case])


def pool_winner_codes(x):
    """Which input of every 2x2 / stride-2 window ATen's max_pool2d picks (row-major scan, a later element replaces the
    maximum only if strictly greater): 0..3 = (dy * 2 + dx)."""
    c = [x[..., dy::2, dx::2] for dy in (0, 1) for dx in (0, 1)]
    best, code = c[0], torch.zeros_like(c[0], dtype=torch.uint8)
    for k in (1, 2, 3):
        upd = c[k] > best
        code = torch.where(upd, torch.full_like(code, k), code)
        best = torch.where(upd, c[k], best)
    return code


def gen_lpips_masks(api):
    """The branch decisions of the reference's float32 LPIPS run behind tests/golden/lpips.npz: one bit per ReLU unit
    (output > 0) of the 13 VGG16 convolutions and two bits per 2x2 max-pool window (the winner), for both images of
    every pair (batch order: in0 then in1).  The trunk and the inputs are the same in both lpips.npz cases, so one set
    serves both.  tests/test_gpu_lpips_masks.py replays the HIP backward with these decisions forced."""
    net = api.LPIPS(net='vgg', lpips=False, pnet_rand=True, pretrained=False, verbose=False)
    torch.nn.Module.load_state_dict(net, cc.det_lpips_state_dict(net), strict=False)
    net.eval()
    in0 = rnd('lpips.in0', (3, 3, 64, 64), 0.5)
    in1 = rnd('lpips.in1', (3, 3, 64, 64), 0.5)
    relu, pool, hooks = {}, {}, []
    for si in range(1, 6):
        for name, mod in getattr(net.net, f'slice{si}').named_children():
            idx = int(name)
            if isinstance(mod, nn.ReLU):            # features[idx - 1] is its convolution
                hooks.append(mod.register_forward_hook(
                    lambda m, i, o, idx=idx: relu.setdefault(idx - 1, []).append((o > 0).clone())))
            elif isinstance(mod, nn.MaxPool2d):
                hooks.append(mod.register_forward_hook(
                    lambda m, i, o, idx=idx: pool.setdefault(idx, []).append(pool_winner_codes(i[0]))))
    with torch.no_grad():
        val = net(in0, in1)
    for h in hooks:
        h.remove()
    ref = np.load(os.path.join(REPO, 'tests', 'golden', 'lpips.npz'))
    assert np.array_equal(val.numpy(), ref['case00/val']), 'not the run stored in lpips.npz'
    case, shapes = {}, {}
    for idx, parts in sorted(relu.items()):
        m = torch.cat(parts, 0).numpy()              # (6, C, H, W): in0's three images, then in1's
        shapes[f'relu{idx}'] = list(m.shape)
        case[f'relu{idx}'] = np.packbits(m.reshape(-1))
    for idx, parts in sorted(pool.items()):
        c = torch.cat(parts, 0).numpy().reshape(-1)
        shapes[f'pool{idx}'] = list(torch.cat(parts, 0).shape)
        c = np.concatenate([c, np.zeros((-c.size) % 4, np.uint8)])
        case[f'pool{idx}'] = (c[0::4] | (c[1::4] << 2) | (c[2::4] << 4) | (c[3::4] << 6)).astype(np.uint8)
    case['meta'] = dict(shapes=shapes, relu_layers=sorted(relu), pool_layers=sorted(pool),
                        active_fraction={str(k): float(torch.cat(v, 0).float().mean()) for k, v in relu.items()})
    save('lpips_masks', [case])


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    api = reference_api()
    only = sys.argv[1:]
    jobs = dict(lpips=lambda: gen_lpips(api), lpips_masks=lambda: gen_lpips_masks(api), act_masks=lambda: gen_act_masks(api), stn_decisions=lambda: gen_stn_decisions(api), c2_generator=lambda: gen_c2_generator(api), c2_stn=lambda: gen_c2_stn(api),
                c1=lambda: gen_config(api, 'c1'), c5=lambda: gen_config(api, 'c5'), c4=lambda: gen_config(api, 'c4'),
                c2=lambda: gen_config(api, 'c2'), c2t=lambda: gen_config(api, 'c2t'), c2r=lambda: gen_config(api, 'c2r'),
                **{n: (lambda n=n: gen_config(api, n)) for n in ('c4b4', 'c5b4', 'c4b16', 'c5b8')},
                c5b16=lambda: gen_config_halves(api, 'c5b16'))
    for name, fn in jobs.items():
        if only and name not in only:
            continue
        t0 = time.time()
        with torch.enable_grad():
            fn()
        print(f'{name}: done in {time.time() - t0:.0f} s')
