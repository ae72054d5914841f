#!/bin/bash
# SQ stall attribution of one conv layer (conv_bench filter in $1); separate --pmc passes, no tracing domains
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GANGEALING_CONV_PRECISION=bf16x3 ITERS=5
F="${1:-G conv 64}"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/p$i -- python $R/scripts/conv_bench.py "$F" > $O/run$i.txt 2>&1
  python $R/scripts/pmc_kernel.py $O/p$i "patch_kernel" > $O/p$i.txt 2>&1
  rm -rf $O/p$i
done
cat $O/p*.txt | cut -c1-200
tail -3 $O/run1.txt
