"""Does any kernel read memory it did not write?  Run the same loss + backward with the allocator's free pool poisoned
with zeros, then with 1e30 / NaN: results must be identical up to atomics noise."""
import os
os.environ.setdefault('GANGEALING_SYNTHETIC', '1')     # random perceptual trunk: synthetic run
import sys; sys.path.insert(0, '/root/repo')
import torch
from gangealing_amd.train_step import GangealingTrainer
from gangealing_amd.op import conv_mfma
dev = torch.device('cuda:0')

def poison(val):
    big = [torch.full((1 << 26,), val, device=dev) for _ in range(6)]          # 6 x 256 MiB
    mid = [torch.full((1 << 18,), val, device=dev) for _ in range(256)]        # 256 x 1 MiB
    small = [torch.full((1 << 8,), val, device=dev) for _ in range(4096)]      # 4096 x 1 KiB
    tiny = [torch.full((1 << 14,), val, device=dev) for _ in range(1024)]      # 64 KiB
    torch.cuda.synchronize()
    del big, mid, small, tiny

for mode in ('fp32', 'bf16x3'):
    conv_mfma.set_precision(mode)
    for cfg in (dict(gen_size=64, flow_size=64, batch=2, inject=3, ndirs=2),
                dict(gen_size=256, flow_size=128, batch=4, inject=5, ndirs=1),
                dict(gen_size=128, flow_size=64, batch=2, inject=6, ndirs=5, num_heads=4, flips=True, sample_from_full_res=True)):
        tr = GangealingTrainer(dev, perturb_heads=0.02, seed=3, **cfg)
        res = []
        for val in (0.0, 1e30, float('nan'), 0.0):
            torch.cuda.empty_cache()
            poison(val)
            torch.manual_seed(7)
            tr.stn_arena.zero_grad(); tr.ll_arena.zero_grad()
            total, parts = tr.loss(0.5)
            with conv_mfma.grad_slots():
                total.backward()
            with torch.no_grad():
                e = tr.t_ema(torch.ones(2, 3, cfg['gen_size'], cfg['gen_size'], device=dev) * 0.1, padding_mode='border')
            torch.cuda.synchronize()
            res.append((float(total), tr.stn_arena.grad.clone(), tr.ll_arena.grad.clone(), e.clone()))
        base = res[0]
        for name, r in zip(('1e30', 'nan', 'zeros-again'), res[1:]):
            gs = float(base[1].abs().max())
            print(mode, cfg['gen_size'], cfg.get('num_heads', 1), name, 'loss', base[0], r[0],
                  'grad diff/max', float((r[1] - base[1]).abs().max()) / gs, 'finite', bool(torch.isfinite(r[1]).all()),
                  'll', float((r[2] - base[2]).abs().max()), 'ema fwd', float((r[3] - base[3]).abs().max()))
