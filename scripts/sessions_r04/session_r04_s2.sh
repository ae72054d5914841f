#!/bin/bash
# Round-4 session: the stride-2 patch tile (conv_s2_patch.hip) against the generic kernel, same box.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_split_precision.py -q -m gpu -x -k "split_conv or s2_patch or block_exponent" 2>&1 | tail -8 > $O/pytest_s2.txt
cat $O/pytest_s2.txt
GG_S2_PATCH=256 timeout 600 python -m pytest tests/test_gpu_split_precision.py -q -m gpu -k "split_conv or s2_patch or block_exponent" 2>&1 | tail -4 > $O/pytest_s2_256.txt
cat $O/pytest_s2_256.txt
for v in 0 128 256; do
  for f in dgrad down; do
    GG_S2_PATCH=$v GANGEALING_CONV_PRECISION=fp16x3 ITERS=20 timeout 300 python scripts/conv_bench.py "$f" 2>&1 | grep -v amdgpu.ids
  done > $O/layers_s2_$v.txt
  echo "== GG_S2_PATCH=$v"; cat $O/layers_s2_$v.txt
done
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5"
for rep in 1 2; do for v in 0 128 256; do
  GG_S2_PATCH=$v $B > $O/bench_s2_${v}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_s2_${v}_$rep.json').read().strip().splitlines() if l.startswith('{')][-1])
print('S2_PATCH=$v rep $rep', d['value'], d['ms_per_step'])
PY
done; done
