#!/bin/bash
# Round 6, session 17: plane-major sign plane of the blur tail (gg_blur4_fused_bits_f32): tests, generator suites, A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s17
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_tail_fusions.py -x -q 2>&1 | tail -12 > $O/pytest_new.txt
cat $O/pytest_new.txt
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_act_masks.py tests/test_gpu_determinism.py tests/test_gpu_prelimb.py tests/test_gpu_poison.py -x -q 2>&1 | tail -8 > $O/pytest_suites.txt
cat $O/pytest_suites.txt
for rep in 1 2 3; do for dis in blur_bits none; do
  GG_DISABLE=$dis python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GG_DISABLE=$dis', d['value'], d['ms_per_step'], d['roofline']['step']['library_calls_per_step'])" >> $O/ab.txt
done; done
cat $O/ab.txt
python scripts/blur_bench.py > $O/blur_bench.txt 2>&1; tail -25 $O/blur_bench.txt
