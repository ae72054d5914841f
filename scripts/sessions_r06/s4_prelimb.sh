#!/bin/bash
# Round 6, session 4: the "activations written once in MFMA-ready form" measurement on the transposed tile.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s4
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
python scripts/prelimb_probe.py > $O/prelimb_times.txt 2>&1
python scripts/prelimb_probe.py >> $O/prelimb_times.txt 2>&1       # twice: same box, order effects
cat $O/prelimb_times.txt
cd /tmp
export ITERS=8 CASE=64
for mode in shipped prelimb; do
  P=$O/pmc_$mode
  MODE=$mode rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $P/a -- python $R/scripts/prelimb_probe.py > /dev/null 2>&1
  MODE=$mode rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $P/b -- python $R/scripts/prelimb_probe.py > /dev/null 2>&1
  MODE=$mode rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $P/c -- python $R/scripts/prelimb_probe.py > /dev/null 2>&1
  for p in a b c; do python $R/scripts/pmc_kernel.py $P/$p "convT3x3s2"; done > $O/pmc_$mode.txt
  rm -rf $P
  cat $O/pmc_$mode.txt
done
