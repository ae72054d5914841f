#!/bin/bash
# Round-4 session: stride-2 weight gradient (conv_s2_wgrad.hip) + the stride-2 patch tile's per-launch tile rule, same box;
# kernel trace of the step in launch order.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2b
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_split_precision.py -q -m gpu 2>&1 | tail -8 > $O/pytest_split.txt
cat $O/pytest_split.txt
timeout 900 python -m pytest tests/test_gpu_c2_layer_ops.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -6 > $O/pytest_c2.txt
cat $O/pytest_c2.txt
for v in 0 1; do
  GG_S2_WGRAD=$v GANGEALING_CONV_PRECISION=fp16x3 ITERS=20 timeout 300 python scripts/conv_bench.py "down" 2>&1 | grep -v amdgpu.ids > $O/layers_wgrad_$v.txt
  echo "== GG_S2_WGRAD=$v"; cat $O/layers_wgrad_$v.txt
done
GANGEALING_CONV_PRECISION=fp16x3 ITERS=20 timeout 300 python scripts/conv_bench.py "dgrad" 2>&1 | grep -v amdgpu.ids > $O/layers_dgrad_rule.txt
cat $O/layers_dgrad_rule.txt
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5"
for rep in 1 2; do for v in 0 1; do
  GG_S2_PATCH=$v GG_S2_WGRAD=$v $B > $O/bench_s2_${v}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_s2_${v}_$rep.json').read().strip().splitlines() if l.startswith('{')][-1])
print('S2 kernels=$v rep $rep', d['value'], d['ms_per_step'])
PY
done; done
timeout 200 python scripts/check_determinism.py c2 fp16x3 2>&1 | tail -1 > $O/determinism.txt; cat $O/determinism.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o trace --output-format rocpd -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_under_rocprofv3.json 2>/dev/null
DB=$(find $O/trace -name "*.db" | head -1)
python $R/scripts/rocpd_stats.py $DB 130 > $O/kernel_stats.txt 2>&1
python $R/scripts/rocpd_timeline.py $DB > $O/timeline.txt 2>&1
rm -rf $O/trace
head -5 $O/timeline.txt
