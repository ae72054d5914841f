#!/bin/bash
# Round-4 session: LPIPS tail kernels (16-byte lanes / 16 channel groups / unrolled channel walk), conv1x1_fewin; same-box A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2d
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_split_precision.py -q -m gpu -k "few_input or s2" 2>&1 | tail -5 > $O/pytest_fewin.txt
cat $O/pytest_fewin.txt
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_lpips_masks.py -q -m gpu 2>&1 | tail -6 > $O/pytest_ops.txt
cat $O/pytest_ops.txt
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_models.py tests/test_gpu_determinism.py -q -m gpu 2>&1 | tail -6 > $O/pytest_cfg.txt
cat $O/pytest_cfg.txt
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5"
run() { # name, env...
  local name=$1; shift
  env "$@" $B > $O/bench_$name.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$name.json').read().strip().splitlines() if l.startswith('{')][-1])
ks={k['kernel']:k for k in d['roofline'].get('kernels',[])} if isinstance(d.get('roofline'),dict) else {}
print('$name', d['value'], d['ms_per_step'])
PY
}
for rep in 1 2; do
  run old_lpips_$rep GANGEALING_HIP_LIB=ab_lib/old_lpips/libgangealing_hip.so
  run new_$rep GG_S2_PATCH=1
  run new_nofewin_$rep GG_NO_FEWIN=1
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o trace --output-format rocpd -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_under_rocprofv3.json 2>/dev/null
DB=$(find $O/trace -name "*.db" | head -1)
python $R/scripts/rocpd_stats.py $DB 130 > $O/kernel_stats.txt 2>&1
python $R/scripts/rocpd_timeline.py $DB > $O/timeline.txt 2>&1
rm -rf $O/trace
grep -i "lpips\|fewin" $O/kernel_stats.txt | cut -c1-150
