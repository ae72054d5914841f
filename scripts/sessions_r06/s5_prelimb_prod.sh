#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s5
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest -q -m gpu -x tests/test_gpu_prelimb.py tests/test_gpu_convt16.py tests/test_gpu_sign_bits.py 2>&1 | tail -30 > $O/pytest.txt
cat $O/pytest.txt
for rep in 1 2; do
  for dis in prelimb none; do
    GG_BENCH_SURVEY_ROWS=40 GG_DISABLE=$dis python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GG_DISABLE=$dis', d['value'], d['ms_per_step'], d['roofline']['step']['mfma_class']['ms_per_step'], d['roofline']['step']['hbm_class']['ms_per_step'], [ (k['kernel'][:44], k['ms_per_step']) for k in d['roofline']['kernels'] if any(t in k['kernel'] for t in ('convT','fewout','torgb','256px,128co,plain'))])" >> $O/ab.txt
  done
done
cat $O/ab.txt
