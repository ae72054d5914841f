"""GPU diagnostic: which torch operators (and from where) does one training iteration launch besides the library's
kernels?  python scripts/op_trace.py [out.txt]"""
import os
import sys

os.environ.setdefault('GANGEALING_SYNTHETIC', '1')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch                                                   # noqa: E402
from torch.profiler import profile, ProfilerActivity          # noqa: E402

from gangealing_amd.op import conv_mfma                        # noqa: E402
from gangealing_amd.train_step import GangealingTrainer        # noqa: E402

conv_mfma.set_precision(os.environ.get('GANGEALING_CONV_PRECISION', 'fp16x3'))
dev = torch.device('cuda', 0)
tr = GangealingTrainer(dev, gen_size=256, flow_size=128, batch=16, transform=('similarity', 'flow'), inject=5, ndirs=1,
                       perturb_heads=0.02, stn_lr=1e-4, ll_lr=1e-4)
for _ in range(3):
    tr.step(psi=0.5)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(psi=0.5)
    torch.cuda.synchronize()
out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else sys.stdout
ka = prof.key_averages(group_by_stack_n=6)
rows = []
for e in ka:
    if not e.key.startswith('aten::'):
        continue
    if e.key in ('aten::empty', 'aten::empty_like', 'aten::view', 'aten::reshape', 'aten::as_strided', 'aten::empty_strided',
                 'aten::detach', 'aten::alias', 'aten::select', 'aten::slice', 'aten::unsqueeze', 'aten::expand', 'aten::t',
                 'aten::transpose', 'aten::permute', 'aten::_unsafe_view', 'aten::squeeze', 'aten::result_type',
                 'aten::is_nonzero', 'aten::item', 'aten::_local_scalar_dense', 'aten::to', 'aten::lift_fresh', 'aten::resolve_conj',
                 'aten::resolve_neg', 'aten::contiguous', 'aten::flatten', 'aten::unbind', 'aten::narrow', 'aten::chunk', 'aten::split'):
        continue
    stack = [s for s in e.stack if 'gangealing_amd' in s or 'bench' in s or 'autograd' in s.lower()][:3]
    rows.append((e.count, e.key, getattr(e, 'device_time_total', 0.0), ' <- '.join(s.split('/')[-1] for s in stack)))
rows.sort(key=lambda r: -r[0])
print(f'{"count":>5s} {"op":28s} {"gpu us":>8s}  where', file=out)
for c, k, t, st in rows[:140]:
    print(f'{c:5d} {k:28s} {t:8.1f}  {st[:150]}', file=out)
