#!/bin/bash
# Round-4 session: row-walking 4x4 / down-2 FIR (upfirdn2d_fir4_down2_kernel); same-box A/B
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2e
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "upfirdn2d or blur or fir" 2>&1 | tail -8 > $O/pytest_fir.txt
cat $O/pytest_fir.txt
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -4 > $O/pytest_cfg.txt
cat $O/pytest_cfg.txt
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5"
run() { local name=$1; shift; env "$@" $B > $O/bench_$name.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$name.json').read().strip().splitlines() if l.startswith('{')][-1])
print('$name', d['value'], d['ms_per_step'])
PY
}
for rep in 1 2 3; do
  run old_fir_$rep GG_NO_FIR4_DOWN2=1
  run new_$rep GG_S2_PATCH=1
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o trace --output-format rocpd -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_under_rocprofv3.json 2>/dev/null
DB=$(find $O/trace -name "*.db" | head -1)
python $R/scripts/rocpd_stats.py $DB 130 > $O/kernel_stats.txt 2>&1
rm -rf $O/trace
grep -i "fir4\|upfirdn2d_direct" $O/kernel_stats.txt | cut -c1-150
