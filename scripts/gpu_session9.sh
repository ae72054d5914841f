#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
python scripts/debug_spike.py 2>&1 | grep -v "Warn\|warn\|amdgpu" | cut -c1-600
