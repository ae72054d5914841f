#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tile
mkdir -p $O
cd $R
export GANGEALING_CONV_PRECISION=bf16x3
for rep in 1 2; do
for v in 0 64 128 256 512; do
  GG_PATCH256_MIN_CIN=$v ITERS=20 python scripts/conv_bench.py conv > $O/bench_${v}_$rep.txt 2>&1
done
done
python - <<'PY'
import re,os
O=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/tile'
rows={}
keys=[(v,r) for r in (1,2) for v in (0,64,128,256,512)]
for k in keys:
    for line in open(f'{O}/bench_{k[0]}_{k[1]}.txt'):
        m=re.match(r'(.*?)\s+\[relerr.*?fwd\s+([\d.]+) ms',line)
        if m: rows.setdefault(m.group(1).strip(),{})[k]=float(m.group(2))
for n,v in rows.items():
    print(f'{n:30s}', ' '.join(f'{k[0]}:{v.get(k,0):6.3f}' for k in keys))
PY
