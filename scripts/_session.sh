#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_ops.py -q -m gpu --maxfail=8 2>&1 | tail -12 > $O/r04g_pytest.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  GANGEALING_OVERLAP_UPDATE=0 $B > $O/r04g_bench_serial_$rep.json 2>/dev/null
  $B > $O/r04g_bench_overlap_$rep.json 2>$O/r04g_bench_overlap_$rep.err
done
python scripts/splat_bench.py $O/r04g_splat_bench.json > $O/r04g_splat_bench.txt 2>&1
for f in $O/r04g_bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([l for l in open('$f').read().strip().splitlines() if l.startswith('{')][-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])
except Exception as e: print('$f', 'ERR', e)"; done
tail -5 $O/r04g_pytest.txt; cut -c1-200 $O/r04g_splat_bench.txt | tail -4
