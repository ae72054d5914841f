#!/bin/bash
# Round-4 session: SQ counters of the transposed 3x3 / stride-2 tile (8-wave variant) on the 512 -> 256 @64^2 -> 129^2 up-convolution
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/convtpmc
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp GANGEALING_CONV_PRECISION=fp16x3 ITERS=8
cd /tmp
CASE="upconv 64"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/a -- python $R/scripts/conv_bench.py "$CASE" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $O/b -- python $R/scripts/conv_bench.py "$CASE" > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/c -- python $R/scripts/conv_bench.py "$CASE" > /dev/null 2>&1
for p in a b c; do python $R/scripts/pmc_kernel.py $O/$p "convT3x3s2" ; rm -rf $O/$p; done > $O/convt.txt
cat $O/convt.txt
python $R/scripts/conv_bench.py "$CASE" | tee $O/layer.txt
