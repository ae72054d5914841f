#!/bin/bash
# Round-4 session: the 16-channel-chunk tile in its stride-1 form (GG_C16_S1=1) against conv3x3_patch_kernel's 128-pixel tile
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c16
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
GG_C16_S1=1 timeout 900 python -m pytest tests/test_gpu_split_precision.py tests/test_gpu_c2_layer_ops.py -q -m gpu 2>&1 | tail -6 > $O/pytest_c16.txt
cat $O/pytest_c16.txt
GG_C16_S1=1 timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_models.py -q -m gpu -x 2>&1 | tail -4 > $O/pytest_cfg.txt
cat $O/pytest_cfg.txt
for v in 0 1; do
  ( for f in "G conv" "STN conv" "STN mask" "VGG"; do GG_C16_S1=$v GANGEALING_CONV_PRECISION=fp16x3 ITERS=20 timeout 300 python scripts/conv_bench.py "$f" 2>&1 | grep -v "amdgpu.ids\|^batch"; done ) > $O/layers_c16_$v.txt
  echo "== GG_C16_S1=$v"; cat $O/layers_c16_$v.txt
done
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5"
run() { local name=$1; shift; env "$@" $B > $O/bench_$name.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$name.json').read().strip().splitlines() if l.startswith('{')][-1])
print('$name', d['value'], d['ms_per_step'])
PY
}
for rep in 1 2; do
  run c16_off_$rep GG_C16_S1=0
  run c16_on_$rep GG_C16_S1=1
done
