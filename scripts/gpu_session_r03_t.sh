#!/bin/bash
# Round-3 session T: SQ counters of the transposed tile (largest up-convolution), shipped loop vs the PIPE2 experiment
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CONVT_ONLY="128->257" ITERS=20 GANGEALING_CONV_PRECISION=fp16x3
for lib in new prev; do
  if [ $lib = prev ]; then export GANGEALING_HIP_LIB=$R/ab_lib/libgg_prevconv.so; fi
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1_$lib -- python $R/scripts/convt_probe.py > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $O/p2_$lib -- python $R/scripts/convt_probe.py > /dev/null 2>&1
  python $R/scripts/pmc_kernel.py $O/p1_$lib convT > $O/counters_$lib.txt 2>&1
  python $R/scripts/pmc_kernel.py $O/p2_$lib convT >> $O/counters_$lib.txt 2>&1
  rm -rf $O/p1_$lib $O/p2_$lib
done
for lib in new prev; do echo "== $lib"; cut -c60-140 $O/counters_$lib.txt; done
