#!/bin/bash
# Round 6, session 14: folded ResBlock + blur-down tap node, similarity-matrix kernel, out-of-place LPIPS sum.
# New tests, the suites they could disturb, then the same-box A/B of the train step (interleaved, 3 repetitions).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s14
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_tail_fusions.py -x -q 2>&1 | tail -15 > $O/pytest_new.txt
cat $O/pytest_new.txt
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_configs.py tests/test_gpu_determinism.py tests/test_gpu_stn_decisions.py tests/test_gpu_act_masks.py -x -q 2>&1 | tail -15 > $O/pytest_suites.txt
cat $O/pytest_suites.txt
for rep in 1 2 3; do for dis in resblock_fold,similarity_matrix none; do
  GG_DISABLE=$dis python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GG_DISABLE=$dis', d['value'], d['ms_per_step'], d['roofline']['step']['library_calls_per_step'])" >> $O/ab.txt
done; done
cat $O/ab.txt
