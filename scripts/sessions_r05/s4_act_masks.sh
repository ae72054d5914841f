#!/bin/bash
# Round 5, session 4: decision-replay tests (generator / STN leaky ReLUs), the tightened gradient floors, LPIPS replay.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
OUT=$R/gpurun_out/r05
mkdir -p $OUT
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_act_masks.py tests/test_gpu_lpips_masks.py -q > $OUT/s4_pytest_masks.txt 2>&1
tail -n 40 $OUT/s4_pytest_masks.txt
python - <<'PY'
import json
d = json.load(open('gpurun_out/parity_report.json'))
for k in d:
    if k.startswith('act_masks'):
        print(k, json.dumps(d[k], indent=1))
PY
cp gpurun_out/parity_report.json $OUT/s4_parity_masks.json
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_reference_dropin.py -q > $OUT/s4_pytest_configs.txt 2>&1
tail -n 8 $OUT/s4_pytest_configs.txt
