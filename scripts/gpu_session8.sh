#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python scripts/debug_graph.py > $O/s8_graph.log 2>&1
grep -v "Warn\|warn\|amdgpu.ids" $O/s8_graph.log | tail -20
