#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -1
B="python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5"
for rep in 1 2; do for b in 1 5; do for v in 1 ""; do
  GG_OLD_STREAM_LOOKUP=$v $B --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('batch $b old_lookup=[$v]', d['value'], d['ms_per_step'])"
done; done; done
python - <<'PY'
# host cost of one library call (a tiny kernel), and of the pieces
import time, torch
from gangealing_amd import _lib
x = torch.zeros(64, device='cuda'); y = torch.empty_like(x)
torch.cuda.synchronize()
def t(f, n=20000):
    t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/n*1e6
print('torch.cuda.current_stream().cuda_stream us', t(lambda: torch.cuda.current_stream().cuda_stream))
print('_cuda_getCurrentRawStream us', t(lambda: torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())))
print('torch add_ launch us', t(lambda: y.add_(1.0)))
from gangealing_amd.op import fused_act
b = torch.zeros(64, device='cuda')
print('fused_leaky_relu (one library call + wrapper) us', t(lambda: fused_act.fused_leaky_relu(x.view(1,64,1,1), b), 5000))
PY
