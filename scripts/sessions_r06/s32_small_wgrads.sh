#!/bin/bash
# Round 6, session 32: weight gradients of the layers the K-slab kernels do not serve (tiny outputs, few output channels)
# on kernels of their own (GG_NO_TINY_WGRAD=1 GG_NO_FEWOUT_WGRAD=1: the exact-fp32 generic tile) - tests, trace A/B, step A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s32
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_tail_fusions.py tests/test_gpu_models.py tests/test_gpu_configs.py tests/test_gpu_determinism.py tests/test_gpu_c2_layer_ops.py tests/test_gpu_stn_decisions.py -q --tb=short 2>&1 | tail -25 > $O/pytest.txt
grep -E "passed|failed|^FAILED|Error" $O/pytest.txt | head -20
cd /tmp
GG_NO_TINY_WGRAD=1 GG_NO_FEWOUT_WGRAD=1 timeout 600 rocprofv3 --kernel-trace -d $O/trace_old -o trace --output-format rocpd -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_old.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace -d $O/trace_new -o trace --output-format rocpd -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_new.json 2>/dev/null
A=$(find $O/trace_old -name "*.db" | head -1); B=$(find $O/trace_new -name "*.db" | head -1)
python $R/scripts/rocpd_diff.py $A $B 15 10 > $O/diff.txt 2>&1
rm -rf $O/trace_old $O/trace_new
cat $O/diff.txt
cd $R
for rep in 1 2 3; do
  GG_NO_TINY_WGRAD=1 GG_NO_FEWOUT_WGRAD=1 python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('generic', d['value'], d['ms_per_step'])" >> $O/ab.txt
  python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
cat $O/ab.txt
