#!/bin/bash
# Round 6, session 16: is cfg_c2t's similarity-stage gradient distance (un-pinned) a function of rounding noise upstream
# of the mip-level arg-max?  The same test under each host-side switch; plus the new stem tests.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s16
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
for dis in resblock_fold,similarity_matrix,vgg_stem resblock_fold,vgg_stem similarity_matrix,vgg_stem resblock_fold,similarity_matrix none; do
  GG_DISABLE=$dis timeout 600 python -m pytest "tests/test_gpu_configs.py::test_config_loss_step" -q -k "c2t or c2-" 2>&1 | tail -3 > $O/pytest_$dis.txt
  python - "$dis" >> $O/summary.txt <<'P'
import json, sys
d = json.load(open('gpurun_out/parity_report.json'))
for k in ('cfg_c2t/similarity-stage', 'cfg_c2/similarity-stage', 'cfg_c2t/flow-stage'):
    v = d.get(k, {})
    print(sys.argv[1], k, {m: round(x['gradients_vs_reference_fp64']['worst_rel_l2_err_ours'], 5) for m, x in v.items()})
P
done
cat $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_tail_fusions.py tests/test_gpu_lpips_masks.py tests/test_gpu_sign_bits.py -x -q 2>&1 | tail -15 > $O/pytest_new.txt
cat $O/pytest_new.txt
for rep in 1 2 3; do for dis in vgg_stem none; do
  GG_DISABLE=$dis python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GG_DISABLE=$dis', d['value'], d['ms_per_step'], d['roofline']['step']['library_calls_per_step'])" >> $O/ab.txt
done; done
cat $O/ab.txt
