#!/bin/bash
# Round 6, session 30: the activation's backward in the Blur's adjoint (gg_blur4_act_bwd_f32, conv1 + act + Blur as one
# node of the folded ResBlock) - tests, per-kernel trace A/B (GG_DISABLE=conv_blur_bwd), train-step A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s30
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_tail_fusions.py tests/test_gpu_models.py tests/test_gpu_configs.py tests/test_gpu_determinism.py tests/test_gpu_poison.py tests/test_gpu_stn_decisions.py tests/test_gpu_act_masks.py tests/test_gpu_ddp.py -q --tb=short 2>&1 | tail -25 > $O/pytest.txt
grep -E "passed|failed|^FAILED|Error" $O/pytest.txt | head -20
cd /tmp
for dis in conv_blur_bwd none; do
  GG_DISABLE=$dis timeout 600 rocprofv3 --kernel-trace -d $O/trace_$dis -o trace --output-format rocpd -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_$dis.json 2>/dev/null
done
A=$(find $O/trace_conv_blur_bwd -name "*.db" | head -1); B=$(find $O/trace_none -name "*.db" | head -1)
python $R/scripts/rocpd_diff.py $A $B 15 10 > $O/diff.txt 2>&1
rm -rf $O/trace_conv_blur_bwd $O/trace_none
cat $O/diff.txt
cd $R
for rep in 1 2 3; do for dis in conv_blur_bwd none; do
  GG_DISABLE=$dis python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GG_DISABLE=$dis', d['value'], d['ms_per_step'])" >> $O/ab.txt
done; done
cat $O/ab.txt
