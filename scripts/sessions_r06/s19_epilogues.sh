#!/bin/bash
# Round 6, session 19: blur sign plane in the kernel's own tiling (no atomics), convolution epilogues (stride-2 conv +
# activation, 1x1 conv + residual), flow-head conv + ReLU.  Tests, per-kernel trace A/B against all of them off, step A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s19
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_tail_fusions.py tests/test_gpu_stn_decisions.py tests/test_gpu_act_masks.py -x -q 2>&1 | tail -12 > $O/pytest_new.txt
cat $O/pytest_new.txt
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_configs.py tests/test_gpu_determinism.py tests/test_gpu_ops.py tests/test_gpu_dropin.py tests/test_gpu_poison.py -q 2>&1 | tail -12 > $O/pytest_suites.txt
cat $O/pytest_suites.txt
OFF=blur_bits,head_relu,conv_s2_act,conv_residual
cd /tmp
for dis in $OFF none; do
  GG_DISABLE=$dis timeout 600 rocprofv3 --kernel-trace -d $O/trace_$dis -o trace --output-format rocpd -- \
    python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$dis.json 2>/dev/null
done
A=$(find $O/trace_$OFF -name "*.db" | head -1); B=$(find $O/trace_none -name "*.db" | head -1)
python $R/scripts/rocpd_diff.py $A $B 15 30 > $O/diff.txt 2>&1
rm -rf $O/trace_$OFF $O/trace_none
cat $O/diff.txt
cd $R
for rep in 1 2 3 4; do for dis in $OFF none; do
  GG_DISABLE=$dis python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GG_DISABLE=$dis', d['value'], d['ms_per_step'], d['roofline']['step']['library_calls_per_step'])" >> $O/ab.txt
done; done
cat $O/ab.txt
