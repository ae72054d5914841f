#!/bin/bash
# Round 5, session 11 (last GPU minutes): (1) the segment-replay reproduction of session 9 ALONE on the GPU, with and
# without the sign plane (is the NaN of the second replay tied to the int32 planes being reused as float buffers?);
# (2) scripts/poison_check.py: every uninitialised torch buffer poisoned - does any kernel read what was never written?
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
OUT=$R/gpurun_out/r05
mkdir -p $OUT
export GG_ENABLE=graph_segments
timeout 70 python scripts/sessions_r05/s9_graph_sync_debug.py flushed 2>&1 | grep "^flushed" | cut -c1-260 > $OUT/s11_segments_default.txt
GG_DISABLE=sign_bits timeout 70 python scripts/sessions_r05/s9_graph_sync_debug.py flushed 2>&1 | grep "^flushed" | cut -c1-260 > $OUT/s11_segments_nosignbits.txt
unset GG_ENABLE
timeout 150 python scripts/poison_check.py > $OUT/s11_poison_check.txt 2>&1
echo "exit $?" >> $OUT/s11_poison_check.txt
tail -n 4 $OUT/s11_segments_default.txt $OUT/s11_segments_nosignbits.txt
grep -v Warning $OUT/s11_poison_check.txt | tail -n 60
