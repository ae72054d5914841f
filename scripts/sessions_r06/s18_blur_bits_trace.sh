#!/bin/bash
# Round 6, session 18: per-kernel A/B of the blur sign plane from two traces on one box (the train-step A/B of session 17
# was inside the box's +-0.2 ms drift); new tests (stem gradient with the kernel's own decisions, flow-head conv + ReLU).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s18
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_tail_fusions.py tests/test_gpu_stn_decisions.py -x -q 2>&1 | tail -12 > $O/pytest_new.txt
cat $O/pytest_new.txt
cd /tmp
for dis in blur_bits none; do
  GG_DISABLE=$dis timeout 600 rocprofv3 --kernel-trace -d $O/trace_$dis -o trace --output-format rocpd -- \
    python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$dis.json 2>/dev/null
done
A=$(find $O/trace_blur_bits -name "*.db" | head -1); B=$(find $O/trace_none -name "*.db" | head -1)
python $R/scripts/rocpd_diff.py $A $B 15 24 > $O/diff.txt 2>&1
python $R/scripts/rocpd_stats.py $B 200 > $O/stats_none.txt 2>&1
rm -rf $O/trace_blur_bits $O/trace_none
cat $O/diff.txt
cd $R
for rep in 1 2 3 4; do for dis in blur_bits,head_relu none; do
  GG_DISABLE=$dis python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GG_DISABLE=$dis', d['value'], d['ms_per_step'], d['roofline']['step']['library_calls_per_step'])" >> $O/ab.txt
done; done
cat $O/ab.txt
