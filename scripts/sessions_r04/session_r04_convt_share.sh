#!/bin/bash
# Round-4 session: the 8-wave transposed tile with an interval's taps ordered by patch offset (patch fragments fetched once per
# offset and k-step: 76 instead of 108 fragment reads per chunk); ab_lib/wbal_before = HEAD before the change
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/convtshare
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_split_precision.py tests/test_gpu_c2_layer_ops.py -q -m gpu -x 2>&1 | tail -4 > $O/pytest.txt
cat $O/pytest.txt
for lib in ab_lib/wbal_before/libgangealing_hip.so gangealing_amd/lib/libgangealing_hip.so; do
  echo "== $lib"
  for rep in 1 2; do GANGEALING_HIP_LIB=$lib GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 timeout 300 python scripts/conv_bench.py "G upconv" 2>&1 | grep -v dgrad | grep "upconv 32\|upconv 64\|upconv 128"; done
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
