#!/bin/bash
# Round-4 session: the launcher's --modules route (this package's modules under the reference's names) beside the
# literal drop-in, tests + bench extras
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/modules
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_reference_dropin.py -q -m gpu 2>&1 | tail -5 > $O/pytest.txt
cat $O/pytest.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_extras.json 2> $O/bench_extras.err
python - <<PY
import json
d=json.loads([l for l in open('$O/bench_extras.json').read().strip().splitlines() if l.startswith('{')][-1])
print('headline', d['value'], d['ms_per_step'])
for k,v in d['extras'].items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('error'))
PY
tail -3 $O/bench_extras.err
