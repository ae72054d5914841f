#!/usr/bin/env python
"""Time the one-launch re-pack of every trainable convolution weight (after the optimizer step) of config C2."""
import os
os.environ.setdefault('GANGEALING_SYNTHETIC', '1')     # random perceptual trunk: synthetic run
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gangealing_amd.op import conv_mfma  # noqa: E402
from gangealing_amd.train_step import GangealingTrainer  # noqa: E402

conv_mfma.set_precision('bf16x3')
tr = GangealingTrainer(torch.device('cuda:0'), gen_size=256, flow_size=128, batch=2, transform=['similarity', 'flow'],
                       perturb_heads=0.02, seed=0)
for _ in range(2):
    tr.step(psi=0.5)
tr.flush()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    conv_mfma.repack_trainable()
s.record()
for _ in range(20):
    conv_mfma.repack_trainable()
e.record()
torch.cuda.synchronize()
reg = conv_mfma.TRAINABLE_PACKS
print(f'jobs {reg.njobs}  repack {s.elapsed_time(e) / 20 * 1e3:.1f} us')
