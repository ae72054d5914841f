#!/bin/bash
# Round 6, session 12: where the batch-independent part of the step is.  (1) torch operators of one iteration with their
# call sites (scripts/op_trace.py); (2) per-kernel time at per-GPU batch 16 vs 32 from two traces on one box
# (a kernel whose time does not double is "fixed" cost); (3) graph-mode bench beside eager on this box.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s12
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 300 python scripts/op_trace.py $O/op_trace.txt > $O/op_trace.log 2>&1
cd /tmp
for b in 16 32; do
  timeout 600 rocprofv3 --kernel-trace -d $O/trace_b$b -o trace --output-format rocpd -- \
    python $R/bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_b$b.json 2>/dev/null
done
A=$(find $O/trace_b16 -name "*.db" | head -1); B=$(find $O/trace_b32 -name "*.db" | head -1)
python $R/scripts/rocpd_diff.py $A $B 15 200 > $O/diff_b16_b32.txt 2>&1
python $R/scripts/rocpd_stats.py $A 200 > $O/stats_b16.txt 2>&1
python $R/scripts/rocpd_timeline.py $A > $O/timeline_b16.txt 2>&1
rm -rf $O/trace_b16 $O/trace_b32
cd $R
python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null > $O/bench_eager.json
python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --graph 2>/dev/null > $O/bench_graph.json
head -c 600 $O/bench_eager.json; echo; head -c 400 $O/bench_graph.json; echo
head -5 $O/diff_b16_b32.txt
