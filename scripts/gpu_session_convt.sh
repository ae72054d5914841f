#!/bin/bash
# A/B of transposed-convolution epilogue variants (ab_lib/ holds the builds) + the per-layer parity tests.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/convt
mkdir -p $O
cd $R
export GANGEALING_CONV_PRECISION=bf16x3
for v in ${VARIANTS:-tpi3 vec noepi}; do
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_$v.so python scripts/conv_bench.py upconv > $O/upconv_$v.txt 2>&1
done
GG_CONVT_TQ64=1 GANGEALING_HIP_LIB=$R/ab_lib/libgg_vec.so python scripts/conv_bench.py upconv > $O/upconv_vec_tq64.txt 2>&1
unset GANGEALING_CONV_PRECISION
timeout 1200 python -m pytest tests/test_gpu_c2_layer_ops.py tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest.txt
for f in $O/upconv_*.txt; do echo "== $f"; cat $f; done
cat $O/pytest.txt
