#!/bin/bash
# Round 6, session 28: planes per thread of upfirdn2d_direct on small planes (GG_DIRECT_BLOCKS4096=1: the former block
# count), the bilinear down-sampler with tabled taps and 32-bit index math - tests, per-kernel trace A/B against the
# committed build of the same sources (ab_lib/stn_old), train-step A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s28
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_tail_fusions.py tests/test_gpu_models.py tests/test_gpu_determinism.py tests/test_gpu_configs.py -x -q 2>&1 | tail -6 > $O/pytest.txt
cat $O/pytest.txt
OLD=$R/ab_lib/lrelu_old/libgangealing_hip.so
cd /tmp
GG_DIRECT_BLOCKS4096=1 GANGEALING_HIP_LIB=$OLD timeout 600 rocprofv3 --kernel-trace -d $O/trace_old -o trace --output-format rocpd -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_old.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace -d $O/trace_new -o trace --output-format rocpd -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_new.json 2>/dev/null
A=$(find $O/trace_old -name "*.db" | head -1); B=$(find $O/trace_new -name "*.db" | head -1)
python $R/scripts/rocpd_diff.py $A $B 15 10 > $O/diff.txt 2>&1
rm -rf $O/trace_old $O/trace_new
cat $O/diff.txt
cd $R
for rep in 1 2 3; do
  GANGEALING_HIP_LIB=$OLD python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('committed-build-before-session-26', d['value'], d['ms_per_step'])" >> $O/ab.txt
  python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
cat $O/ab.txt
