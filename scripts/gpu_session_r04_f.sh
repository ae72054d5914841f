#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  GANGEALING_F16_GRADS=0 GANGEALING_HIP_LIB=$PWD/ab_lib/r03conv/libgangealing_hip.so $B > $O/r04f_bench_r03conv_$rep.json 2>/dev/null
  $B > $O/r04f_bench_cur_$rep.json 2>/dev/null
  for v in mtop alltop t8; do
    GANGEALING_HIP_LIB=$PWD/ab_lib/$v/libgangealing_hip.so $B > $O/r04f_bench_${v}_$rep.json 2>/dev/null
  done
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_splat -o trace --output-format rocpd -- python $GRAFT_REPO_ROOT/scripts/splat_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/trace_splat -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB 30 > $O/r04f_splat_kernel_stats.txt 2>&1
rm -rf $O/trace_splat
for f in $O/r04f_bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([l for l in open('$f').read().strip().splitlines() if l.startswith('{')][-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])
except Exception as e: print('$f', 'ERR', e)"; done
head -20 $O/r04f_splat_kernel_stats.txt | cut -c1-160
