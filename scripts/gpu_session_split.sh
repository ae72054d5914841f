#!/bin/bash
# split-K threshold sweep of the patch kernels on the per-layer benchmark
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/split
mkdir -p $O
cd $R
export GANGEALING_CONV_PRECISION=bf16x3
for t in 512 256 128 1; do
  GG_SPLIT_PATCH=$t GG_SPLIT_CONVT=$t python scripts/conv_bench.py > $O/bench_$t.txt 2>&1
done
python - <<'PY'
import re,os
O=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/split'
rows={}
for t in (512,256,128,1):
    for line in open(f'{O}/bench_{t}.txt'):
        m=re.match(r'(.*?)\s+\[relerr.*?fwd\s+([\d.]+) ms',line)
        if m: rows.setdefault(m.group(1).strip(),{})[t]=float(m.group(2))
        m2=re.search(r'wgrad\s+([\d.]+) ms',line)
for k,v in rows.items():
    print(f'{k:36s}', '  '.join(f'{t}:{v.get(t,0):7.3f}' for t in (512,256,128,1)))
PY
