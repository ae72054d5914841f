#!/bin/bash
# per-layer A/B of library builds: gpu_session_layers.sh "<filter>" libA libB ...   (names under ab_lib/ without libgg_/.so)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/layers
mkdir -p $O
cd $R
export GANGEALING_CONV_PRECISION=${PREC:-bf16x3} ITERS=20
F="$1"; shift
for rep in 1 2; do
for v in "$@"; do
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_$v.so python scripts/conv_bench.py $F > $O/bench_${v}_$rep.txt 2>&1
done
done
V="$*" python - <<'PY'
import re,os
O=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/layers'
vs=os.environ['V'].split()
rows={}
keys=[(v,r) for r in (1,2) for v in vs]
for k in keys:
    for line in open(f'{O}/bench_{k[0]}_{k[1]}.txt'):
        m=re.match(r'(.*?)\s+\[relerr ([\d.e+-]+)\].*?fwd\s+([\d.]+) ms',line)
        if m: rows.setdefault(m.group(1).strip(),{})[k]=(float(m.group(3)),m.group(2))
for n,v in rows.items():
    print(f'{n:30s}', ' '.join(f'{k[0]}:{v.get(k,(0,0))[0]:6.3f}' for k in keys), ' err', v.get(keys[-1],(0,'?'))[1])
PY
if [ -n "${TESTS:-}" ]; then timeout 1500 python -m pytest $TESTS -m gpu -q -x 2>&1 | tail -5; fi
