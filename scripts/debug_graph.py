import os
os.environ.setdefault('GANGEALING_SYNTHETIC', '1')     # random perceptual trunk: synthetic run
import sys; sys.path.insert(0, '/root/repo')
import torch
from gangealing_amd.train_step import GangealingTrainer
from gangealing_amd.op import conv_mfma
dev = torch.device('cuda:0')
conv_mfma.set_precision('bf16x3')
for sync in (True, False):
    cfg = dict(gen_size=256, flow_size=128, batch=16)
    tr = GangealingTrainer(dev, perturb_heads=0.02, seed=0, use_graph=True, **cfg)
    out = []
    for i in range(30):
        parts = tr.step(psi=0.5)
        if sync:
            torch.cuda.synchronize()
        out.append((parts['p'].clone(), parts['tv'].clone(), tr.stn_arena.param.abs().max().clone(), tr.stn_arena.grad.abs().max().clone()))
    torch.cuda.synchronize()
    print('sync' if sync else 'nosync', [tuple(round(float(v), 5) for v in o) for o in out], flush=True)
    del tr
    torch.cuda.empty_cache()
# the eager reference of the same run
tr = GangealingTrainer(dev, perturb_heads=0.02, seed=0, use_graph=False, **cfg)
out = []
for i in range(30):
    parts = tr.step(psi=0.5)
    out.append((parts['p'].clone(), parts['tv'].clone(), tr.stn_arena.param.abs().max().clone(), tr.stn_arena.grad.abs().max().clone()))
torch.cuda.synchronize()
print('eager', [tuple(round(float(v), 5) for v in o) for o in out], flush=True)
