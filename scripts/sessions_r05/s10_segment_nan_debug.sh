#!/bin/bash
# Round 5, session 10: bisect the NaN of the second segment replay (un-forked generator passes) with the library's
# run-time switches; all variants run side by side on the one GPU (tiny configuration).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
OUT=$R/gpurun_out/r05
mkdir -p $OUT
S=scripts/sessions_r05/s10_segment_nan_debug.py
python -c "import torch" 2>/dev/null        # page the image in once
run() { # label, extra args, env...
  local label=$1; shift; local extra=$1; shift
  ( env "$@" timeout 200 python $S $label $extra > $OUT/s10_$label.txt 2>&1; echo "exit $?" >> $OUT/s10_$label.txt ) &
}
run baseline "" X=1
run plain "" S10_NOSTASH=1
run fork "" GG_ENABLE=two_streams
run nopackreg "" GG_DISABLE=pack_registry
run nosignbits "" GG_DISABLE=sign_bits
run nobanks "" GG_DISABLE=style_bank,noise_bank
run noslots "" GG_DISABLE=slots
run single "single" X=1
run single_fork "single" GG_ENABLE=two_streams
wait
for f in $OUT/s10_*.txt; do echo "== $f"; grep -v Warning $f | cut -c1-400 | tail -n 30; done
