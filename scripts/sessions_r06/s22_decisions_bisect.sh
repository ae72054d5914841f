#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s22
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
for dis in none resblock_fold similarity_matrix conv_residual head_relu blur_bits vgg_stem; do
  GG_DISABLE=$dis timeout 300 python -m pytest "tests/test_gpu_stn_decisions.py" -q -k fp32 2>&1 | tail -1 > $O/out_$dis.txt
  python - "$dis" >> $O/summary.txt <<'P'
import json, sys
d = json.load(open('gpurun_out/parity_report.json'))['stn_decisions']['fp32']
print(sys.argv[1], 'free', round(d['free']['similarity_stage_worst_rel_l2_vs_reference_fp32'], 7), d['free']['similarity_stage_worst_param'],
      'pinned', round(d['pinned']['similarity_stage_worst_rel_l2_vs_reference_fp32'], 7), d['pinned']['similarity_stage_worst_param'])
P
done
cat $O/summary.txt
