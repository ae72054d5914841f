"""GPU diagnostic: every torch operator one training iteration dispatches (forward AND backward), with the tensor shapes
and the innermost gangealing_amd call sites - a TorchDispatchMode, so it also sees the operators autograd's own nodes
issue (no Python frame: marked <autograd>).  python scripts/aten_sources.py [out.txt]"""
import collections
import os
import sys
import traceback

os.environ.setdefault('GANGEALING_SYNTHETIC', '1')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch                                                   # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode     # noqa: E402

from gangealing_amd.op import conv_mfma                        # noqa: E402
from gangealing_amd.train_step import GangealingTrainer        # noqa: E402

SKIP = {'view', 'reshape', 'as_strided', 'detach', 'alias', 'select', 'slice', 'unsqueeze', 'expand', 't', 'transpose',
        'permute', '_unsafe_view', 'squeeze', 'unbind', 'empty', 'empty_like', 'empty_strided', 'new_empty',
        'new_empty_strided', 'lift_fresh', 'unfold', 'split', 'narrow', '_reshape_alias', 'resize_', 'view_as', 'expand_as',
        'is_nonzero', 'item', '_local_scalar_dense', 'is_pinned', 'set_', 'record_stream'}


class Tracer(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in SKIP:
            shapes = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))[:3]
            frames = [f for f in traceback.extract_stack() if 'gangealing_amd' in f.filename and 'aten_sources' not in f.filename]
            where = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(frames[-3:])) or '<autograd>'
            self.rows[(name, shapes, where)] += 1
        return func(*args, **(kwargs or {}))


conv_mfma.set_precision(os.environ.get('GANGEALING_CONV_PRECISION', 'fp16x3'))
dev = torch.device('cuda', 0)
tr = GangealingTrainer(dev, gen_size=256, flow_size=128, batch=16, transform=('similarity', 'flow'), inject=5, ndirs=1,
                       perturb_heads=0.02, stn_lr=1e-4, ll_lr=1e-4)
for _ in range(3):
    tr.step(psi=0.5)
torch.cuda.synchronize()
t = Tracer()
with t:
    tr.step(psi=0.5)
torch.cuda.synchronize()
out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else sys.stdout
print(f'# {sum(t.rows.values())} dispatched operators (views / allocations not listed)', file=out)
for (name, shapes, where), c in sorted(t.rows.items(), key=lambda kv: (kv[0][2], kv[0][0])):
    print(f'{c:4d} {name:22s} {str(shapes)[:70]:70s} {where}', file=out)
