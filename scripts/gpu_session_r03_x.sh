#!/bin/bash
# Round-3 session X: cache policy of the transposed tile's epilogue stores (buffer_store aux bits: 1 = sc0, 2 = nt, 3 = sc0 nt)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03x
mkdir -p $O
cd $R
export CONVT_ONLY="upconv" GANGEALING_CONV_PRECISION=fp16x3
python scripts/convt_probe.py > $O/probe_aux0.txt 2>&1
for v in 1 2 3; do
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_aux$v.so python scripts/convt_probe.py > $O/probe_aux$v.txt 2>&1
done
GANGEALING_HIP_LIB=$R/ab_lib/libgg_aux2.so ITERS=30 python scripts/conv_bench.py "G conv" > $O/gconv_aux2.txt 2>&1
ITERS=30 python scripts/conv_bench.py "G conv" > $O/gconv_aux0.txt 2>&1
paste <(grep upconv $O/probe_aux0.txt) <(grep upconv $O/probe_aux1.txt | awk '{print $(NF-3)}') <(grep upconv $O/probe_aux2.txt | awk '{print $(NF-3)}') <(grep upconv $O/probe_aux3.txt | awk '{print $(NF-3)}')
paste <(grep "G conv" $O/gconv_aux0.txt | cut -c1-100) <(grep "G conv" $O/gconv_aux2.txt | cut -c66-100)
