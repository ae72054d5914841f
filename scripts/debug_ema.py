import os
os.environ.setdefault('GANGEALING_SYNTHETIC', '1')     # random perceptual trunk: synthetic run
import sys; sys.path.insert(0, '/root/repo')
import torch
from gangealing_amd.train_step import GangealingTrainer
from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
from gangealing_amd.op import conv_mfma
from gangealing_amd.stylegan2.networks import EqualLinear
cuda = torch.device('cuda:0')
kw = dict(gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2, perturb_heads=0.05, seed=21)
tr = GangealingTrainer(cuda, stn_lr=2e-2, **kw)
tr.ema_decay = 0.0
x = torch.randn(2, 3, 64, 64, device=cuda) * 0.5
with torch.no_grad():
    out0, flow0 = tr.t_ema(x, return_flow=True, padding_mode='reflection')
for it in range(2):
    tr.step(psi=0.5); tr.flush()
    print('ema == stn params:', float((tr.ema_arena.param - tr.stn_arena.param).abs().max()))
    with torch.no_grad():
        o1, f1 = tr.t_ema(x, return_flow=True, padding_mode='reflection')
        o1b, f1b = tr.t_ema(x, return_flow=True, padding_mode='reflection')
        conv_mfma._FROZEN_PACKS.clear()
        for m in tr.t_ema.modules():
            if isinstance(m, EqualLinear): m._scaled_cache = None
        o2, f2 = tr.t_ema(x, return_flow=True, padding_mode='reflection')
        fresh = get_stn(['similarity', 'flow'], flow_size=64, supersize=64, channel_multiplier=0.5, num_heads=1).to(cuda)
        fresh.load_state_dict(tr.t_ema.state_dict()); fresh.requires_grad_(False)
        o3, f3 = fresh(x, return_flow=True, padding_mode='reflection')
        o4, f4 = tr.stn(x, return_flow=True, padding_mode='reflection')
    d = lambda a, b: float((a - b).abs().max())
    print(it, 'vs prev', d(f1, flow0), 'repeat', d(f1, f1b), 'after cache clear', d(f1, f2), 'fresh vs ema', d(f3, f1), 'fresh vs cleared', d(f3, f2), 'stn vs fresh', d(f4, f3))
    for (n1, p1), (n2, p2) in zip(tr.t_ema.named_parameters(), fresh.named_parameters()):
        if d(p1, p2) > 0: print('param differs', n1, d(p1, p2))
    flow0 = f1
