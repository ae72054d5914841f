#!/bin/bash
# Round-4 closing A/B on one box: the library of commit c034090 (end of the round's first half: block exponents, splat2d
# gather, ...) with the old style path against HEAD, train step C2 / C4 / C5
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final_ab
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
OLD="GANGEALING_HIP_LIB=ab_lib/round4_first_half/libgangealing_hip.so GG_DISABLE=style_demod_grad GG_S2_PATCH=0"
run() { local name=$1; shift; env "$@" > $O/bench_$name.json 2>/dev/null
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$name.json').read().strip().splitlines() if l.startswith('{')][-1])
print('$name', d['value'], d['ms_per_step'])
PY
}
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5"
for rep in 1 2 3; do
  run c2_old_$rep $OLD $B
  run c2_new_$rep GG_S2_PATCH=1 $B
done
for w in c4 c5; do
  run ${w}_old $OLD $B --workload $w --batch 16 --steps 10 --warmup 3
  run ${w}_new GG_S2_PATCH=1 $B --workload $w --batch 16 --steps 10 --warmup 3
done
run c2b5_old $OLD $B --batch 5
run c2b5_new GG_S2_PATCH=1 $B --batch 5
