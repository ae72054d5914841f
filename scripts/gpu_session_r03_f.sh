#!/bin/bash
# Round-3 session F: split-K thresholds of the patch kernels now that a split launch costs a reduce pass instead of a
# memset (GG_SPLIT_PATCH / GG_SPLIT_CONVT: blocks below which a launch is split along Cin), at batch 16 and 5.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03f
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "torch_library or wgrad" 2>&1 | tail -2 > $O/pytest.txt
for b in 16 5; do
  for sp in 512 256 128 1; do
    GG_SPLIT_PATCH=$sp python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_b${b}_patch$sp.json 2>/dev/null
  done
  for sc in 192 96 1 384; do
    GG_SPLIT_CONVT=$sc python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_b${b}_convt$sc.json 2>/dev/null
  done
done
cat $O/pytest.txt
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])"; done
