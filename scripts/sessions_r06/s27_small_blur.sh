#!/bin/bash
# Round 6, session 27: 4x4 blur on small planes (<= 16^2) on the constant-trip-count path of upfirdn2d_direct
# (GG_NO_FIR4_SMALL=1: the generic loop) - tests, per-kernel trace A/B, train-step A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s27
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_tail_fusions.py tests/test_gpu_models.py tests/test_gpu_determinism.py -x -q 2>&1 | tail -6 > $O/pytest.txt
cat $O/pytest.txt
cd /tmp
for v in 1 0; do
  if [ $v = 1 ]; then export GG_NO_FIR4_SMALL=1; else unset GG_NO_FIR4_SMALL; fi
  timeout 600 rocprofv3 --kernel-trace -d $O/trace_$v -o trace --output-format rocpd -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$v.json 2>/dev/null
done
unset GG_NO_FIR4_SMALL
A=$(find $O/trace_1 -name "*.db" | head -1); B=$(find $O/trace_0 -name "*.db" | head -1)
python $R/scripts/rocpd_diff.py $A $B 15 6 > $O/diff.txt 2>&1
rm -rf $O/trace_0 $O/trace_1
cat $O/diff.txt
cd $R
for rep in 1 2 3; do
  GG_NO_FIR4_SMALL=1 python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('generic', d['value'], d['ms_per_step'])" >> $O/ab.txt
  python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fast', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
cat $O/ab.txt
