#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s6
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd /tmp
for dis in prelimb none; do
  GG_DISABLE=$dis timeout 600 rocprofv3 --kernel-trace -d $O/trace_$dis -o trace --output-format rocpd -- \
    python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$dis.json 2>/dev/null
done
A=$(find $O/trace_prelimb -name "*.db" | head -1); B=$(find $O/trace_none -name "*.db" | head -1)
python $R/scripts/rocpd_diff.py $A $B 15 30 > $O/diff.txt 2>&1
rm -rf $O/trace_prelimb $O/trace_none
cat $O/diff.txt
cd $R
for rep in 1 2 3; do for dis in prelimb none; do
  GG_DISABLE=$dis python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GG_DISABLE=$dis', d['value'], d['ms_per_step'])" >> $O/ab.txt
done; done
cat $O/ab.txt
for w in c4 c5; do for dis in prelimb none; do
  GG_DISABLE=$dis python bench.py --workload $w --batch 16 --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w GG_DISABLE=$dis', d['value'], d['ms_per_step'])" >> $O/ab_c45.txt
done; done
cat $O/ab_c45.txt
