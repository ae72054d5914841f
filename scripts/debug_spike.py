import os
os.environ.setdefault('GANGEALING_SYNTHETIC', '1')     # random perceptual trunk: synthetic run
import sys; sys.path.insert(0, '/root/repo')
import torch
from gangealing_amd.train_step import GangealingTrainer
from gangealing_amd.op import conv_mfma
import gangealing_amd.spatial_transformers.warping_heads as wh
import gangealing_amd.spatial_transformers.antialiased_sampling as aa
dev = torch.device('cuda:0')
conv_mfma.set_precision('bf16x3')
log = []
def hook(name):
    def h(g):
        log.append((name, float(g.abs().max()), tuple(g.shape)))
    return h
orig_affine, orig_compose = wh.affine_grid, wh.flow_compose
def affine_grid(matrix, size):
    if matrix.requires_grad: matrix.register_hook(hook('sim.matrix'))
    g = orig_affine(matrix, size)
    if g.requires_grad: g.register_hook(hook('sim.grid'))
    return g
def flow_compose(low, mask, base, ds):
    if base is not None and base.requires_grad: base.register_hook(hook('flow.base_warp'))
    if low.requires_grad: low.register_hook(hook('flow.low'))
    flow, delta = orig_compose(low, mask, base, ds)
    flow.register_hook(hook('flow.flow')); delta.register_hook(hook('flow.delta'))
    return flow, delta
wh.affine_grid, wh.flow_compose = affine_grid, flow_compose
orig_fn = aa._MipmapWarpFn.backward
def bwd(ctx, grad_out, gl):
    res = orig_fn(ctx, grad_out, gl)
    grid = ctx.saved_tensors[0]
    log.append(('warp.bwd', float(grad_out.abs().max()), float(res[1].abs().max()), tuple(grid.shape), ctx.conf[9:]))
    return res
aa._MipmapWarpFn.backward = staticmethod(bwd)
tr = GangealingTrainer(dev, perturb_heads=0.02, seed=0, gen_size=256, flow_size=128, batch=16)
for i in range(9):
    log.clear()
    parts = tr.step(psi=0.5)
    torch.cuda.synchronize()
    g = max(float(p.grad.abs().max()) for p in tr.stn.parameters())
    lv = tr.stn.stns[0].warp_head.warper.levels_map
    lv2 = tr.stn.stns[1].warp_head.warper.levels_map
    print(i, 'max grad', round(g, 3), 'sim levels min/max', float(lv.min()) * 2.5, float(lv.max()) * 2.5, 'flow levels', float(lv2.min()) * 2.5, float(lv2.max()) * 2.5)
    if g > 100 or i == 0:
        for e in log: print('    ', e)
