#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s3
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 1700 python -m pytest -q -m gpu tests/test_gpu_stn_decisions.py tests/test_gpu_config_full_tensors.py tests/test_gpu_ops.py tests/test_gpu_indices.py tests/test_gpu_act_masks.py 2>&1 | tail -40 > $O/pytest.txt
cat $O/pytest.txt
python - <<'PY'
import json
d = json.load(open('gpurun_out/parity_report.json'))
print(json.dumps(d.get('stn_decisions'), indent=1))
print(json.dumps(d.get('cfg_c2_full'), indent=1))
PY
cp gpurun_out/parity_report.json $O/parity_partial.json
