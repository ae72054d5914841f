#!/bin/bash
# Round-4 session: the 8-wave stride-2 tile's weight slabs moved by all eight waves (ab_lib/wbal_before = by the first four)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/wbal
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
GG_S2_PATCH=256 timeout 600 python -m pytest tests/test_gpu_split_precision.py -q -m gpu -k "split_conv or s2_patch or block_exponent" 2>&1 | tail -3 > $O/pytest_256.txt
cat $O/pytest_256.txt
for lib in ab_lib/wbal_before/libgangealing_hip.so gangealing_amd/lib/libgangealing_hip.so; do
  echo "== $lib"
  for rep in 1 2; do GANGEALING_HIP_LIB=$lib GG_S2_PATCH=256 GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 timeout 300 python scripts/conv_bench.py "dgrad" 2>&1 | grep "dgrad 65\|dgrad 129\|dgrad 257"; done
done > $O/layers.txt
cat $O/layers.txt
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5"
run() { local name=$1; shift; env "$@" $B > $O/bench_$name.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$name.json').read().strip().splitlines() if l.startswith('{')][-1])
print('$name', d['value'], d['ms_per_step'])
PY
}
for rep in 1 2; do
  run before_$rep GANGEALING_HIP_LIB=ab_lib/wbal_before/libgangealing_hip.so
  run after_$rep GG_S2_PATCH=1
done
