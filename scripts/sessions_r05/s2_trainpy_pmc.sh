#!/bin/bash
# Round 5, session 2: train.py itself under the launcher (tests), SQ counters of the 16-channel-chunk transposed tile,
# the full bench line with the C4 / C5 extras.
#   gpurun --timeout 1500 -- 'bash scripts/sessions_r05/s2_trainpy_pmc.sh'
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
OUT=$R/gpurun_out/r05
mkdir -p $OUT
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train_script.py tests/test_gpu_convt16.py -q -x > $OUT/s2_pytest.txt 2>&1
tail -n 30 $OUT/s2_pytest.txt
# SQ counters: 512 -> 256 @64^2 -> 129^2 (batch 16, fp16x3) on the new tile (64 co x 128 q, two blocks per CU) and, same
# box, the round-4 tile
export GANGEALING_CONV_PRECISION=fp16x3 ITERS=8
cd /tmp
CASE="upconv 64"
for mode in 64 0; do
  O=$OUT/pmc_mode$mode
  mkdir -p $O
  GG_CONVT16=$mode rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/a -- python $R/scripts/conv_bench.py "$CASE" > /dev/null 2>&1
  GG_CONVT16=$mode rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $O/b -- python $R/scripts/conv_bench.py "$CASE" > /dev/null 2>&1
  GG_CONVT16=$mode rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/c -- python $R/scripts/conv_bench.py "$CASE" > /dev/null 2>&1
  for p in a b c; do python $R/scripts/pmc_kernel.py $O/$p "convT3x3s2"; done > $OUT/s2_pmc_mode$mode.txt
  rm -rf $O
  GG_CONVT16=$mode python $R/scripts/conv_bench.py "$CASE" >> $OUT/s2_pmc_mode$mode.txt 2>&1
  cat $OUT/s2_pmc_mode$mode.txt
done
cd $R
unset GANGEALING_CONV_PRECISION ITERS
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/s2_bench_full.json 2> $OUT/s2_bench_full.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05/s2_bench_full.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
for k, v in d.get('extras', {}).items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'error')}, (v.get('roofline') or {}).get('kernel'), (v.get('roofline') or {}).get('frac'))
print(d.get('cpu_baseline'))
PY
