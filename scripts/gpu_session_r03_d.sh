#!/bin/bash
# Round-3 session D: full GPU suite after the tail changes (style-gradient kernel, bias slots, 3x3 few-output kernel,
# skip-path blur at stride), bench A/B against the session-A library, kernel trace.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03d
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
for rep in 1 2; do
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_r03a.so GG_DISABLE=skip_down,style_grad python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_old_$rep.json 2>$O/err_old_$rep.txt
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_new_$rep.json 2>$O/err_new_$rep.txt
done
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o trace --output-format rocpd -- $CMD > $O/bench_under_rocprofv3.json 2>/dev/null
DB=$(find $O/trace -name "*.db" | head -1)
python $R/scripts/rocpd_stats.py $DB 120 > $O/kernel_stats.txt 2>&1
rm -rf $O/trace
cd $R
tail -6 $O/pytest.log
for f in $O/bench_*_?.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
head -3 $O/kernel_stats.txt | cut -c1-150
