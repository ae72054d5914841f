#!/bin/bash
# Round-4 session: stride-2 weight gradient fixed + style_demod_grad; wider test subset; same-box A/B of the three changes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2c
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_split_precision.py -q -m gpu 2>&1 | tail -12 > $O/pytest_split.txt
cat $O/pytest_split.txt
timeout 1200 python -m pytest tests/test_gpu_c2_layer_ops.py tests/test_gpu_configs.py tests/test_gpu_models.py tests/test_gpu_dropin.py -q -m gpu 2>&1 | tail -12 > $O/pytest_c2.txt
cat $O/pytest_c2.txt
for v in 0 1; do
  GG_S2_WGRAD=$v GANGEALING_CONV_PRECISION=fp16x3 ITERS=20 timeout 300 python scripts/conv_bench.py "down" 2>&1 | grep -v amdgpu.ids > $O/layers_wgrad_$v.txt
  echo "== GG_S2_WGRAD=$v"; cat $O/layers_wgrad_$v.txt
done
GANGEALING_CONV_PRECISION=fp16x3 ITERS=20 timeout 300 python scripts/conv_bench.py "dgrad" 2>&1 | grep -v amdgpu.ids > $O/layers_dgrad_rule.txt
cat $O/layers_dgrad_rule.txt
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5"
run() { # name, env...
  local name=$1; shift
  env "$@" $B > $O/bench_$name.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$name.json').read().strip().splitlines() if l.startswith('{')][-1])
print('$name', d['value'], d['ms_per_step'])
PY
}
for rep in 1 2; do
  run old_$rep GG_S2_PATCH=0 GG_S2_WGRAD=0 GG_DISABLE=style_demod_grad
  run new_$rep GG_S2_PATCH=1
  run new_nostyle_$rep GG_DISABLE=style_demod_grad
  run new_nowgrad_$rep GG_S2_WGRAD=0
done
timeout 200 python scripts/check_determinism.py c2 fp16x3 2>&1 | tail -1 > $O/determinism.txt; cat $O/determinism.txt
