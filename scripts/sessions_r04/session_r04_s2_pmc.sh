#!/bin/bash
# Round-4 session: SQ counters of the stride-2 patch tile (both variants) on the 256 -> 512 @129^2 -> 64^2 data gradient
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/s2pmc
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp GANGEALING_CONV_PRECISION=fp16x3 ITERS=8
cd /tmp
for v in 256 128; do
  CMD="python $R/scripts/conv_bench.py dgrad-129"
  GG_S2_PATCH=$v rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/a_$v -- python $R/scripts/conv_bench.py "dgrad 129" > /dev/null 2>&1
  GG_S2_PATCH=$v rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $O/b_$v -- python $R/scripts/conv_bench.py "dgrad 129" > /dev/null 2>&1
  GG_S2_PATCH=$v rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/c_$v -- python $R/scripts/conv_bench.py "dgrad 129" > /dev/null 2>&1
  for p in a b c; do python $R/scripts/pmc_kernel.py $O/${p}_$v "conv3x3s2" ; rm -rf $O/${p}_$v; done > $O/s2_$v.txt
  echo "== tile $v"; cat $O/s2_$v.txt
done
