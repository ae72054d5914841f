#!/bin/bash
# Build a second copy of the library whose conv_mfma.hip comes from another commit (same-box A/B of kernel changes):
#   scripts/build_ab_lib.sh <commit | path/to/conv_mfma.hip> <name>   ->  ab_lib/<name>/libgangealing_hip.so
#   (git-ignored; travels with gpurun)
# Everything else (runtime, ABI number, the other kernels) is today's, so the current Python side loads it:
#   GANGEALING_HIP_LIB=ab_lib/<name>/libgangealing_hip.so python bench.py ...
set -eu
cd "$(dirname "$0")/.."
COMMIT=$1; NAME=$2
OUT=ab_lib/$NAME; mkdir -p $OUT/build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result"
mkdir -p $OUT/src/gangealing_amd/csrc $OUT/src/include
cp gangealing_amd/csrc/*.hip gangealing_amd/csrc/gg_common.h $OUT/src/gangealing_amd/csrc/
cp include/gangealing_hip.h $OUT/src/include/
if [ -f "$COMMIT" ]; then cp "$COMMIT" $OUT/src/gangealing_amd/csrc/conv_mfma.hip
else git show $COMMIT:gangealing_amd/csrc/conv_mfma.hip > $OUT/src/gangealing_amd/csrc/conv_mfma.hip; fi
( cd $OUT/src/gangealing_amd/csrc && $HIPCC $FLAGS -c conv_mfma.hip -o ../../../build/conv_mfma.o )
OBJS="$OUT/build/conv_mfma.o"
for f in gg_runtime fused_bias_act upfirdn2d splat2d mipmap_warp stn_ops modulation lpips optim; do OBJS="$OBJS gangealing_amd/csrc/build/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libgangealing_hip.so $OBJS
rm -rf $OUT/src $OUT/build
ls -la $OUT
