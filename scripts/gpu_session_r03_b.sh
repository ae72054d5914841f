#!/bin/bash
# Round-3 session B: rocprofv3 kernel traces of the bench command for c2 / c4 / c5 (what dominates where).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03b
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd /tmp && export TMPDIR=/tmp
for w in c2 c4 c5; do
  CMD="python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
  timeout 600 rocprofv3 --kernel-trace -d $O/trace_$w -o trace --output-format rocpd -- $CMD > $O/bench_${w}_under_rocprofv3.json 2>/dev/null
  DB=$(find $O/trace_$w -name "*.db" | head -1)
  python $R/scripts/rocpd_stats.py $DB 110 > $O/kernel_stats_$w.txt 2>&1
  rm -rf $O/trace_$w
done
head -30 $O/kernel_stats_c2.txt | cut -c1-180
