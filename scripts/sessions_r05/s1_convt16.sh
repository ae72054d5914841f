#!/bin/bash
# Round 5, session 1: the 16-channel-chunk transposed tile (conv_t_c16.hip) against the round-4 tile on one box.
#   gpurun --timeout 1500 -- 'bash scripts/sessions_r05/s1_convt16.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05
mkdir -p $OUT
export GANGEALING_CONV_PRECISION=fp16x3
python -c "import torch; print(torch.cuda.get_device_name(0))" > $OUT/s1_device.txt 2>&1
# parity of the new tile first (forced through gg_set_tuning in both forms)
timeout 900 python -m pytest tests/test_gpu_convt16.py -x -q > $OUT/s1_pytest_convt16.txt 2>&1
tail -n 5 $OUT/s1_pytest_convt16.txt
# per-layer A/B: the generator's up-convolutions and the STN's stride-2 data gradient (batch 16)
for mode in 0 64 128 0 64; do
  echo "== GG_CONVT16=$mode" >> $OUT/s1_convt16_layers.txt
  GG_CONVT16=$mode ITERS=20 timeout 300 python scripts/conv_bench.py 'upconv ' >> $OUT/s1_convt16_layers.txt 2>&1
  GG_CONVT16=$mode ITERS=20 timeout 300 python scripts/conv_bench.py 'down dgrad' >> $OUT/s1_convt16_layers.txt 2>&1
done
echo "== GG_CONVT16=64 GG_CONVT16_TW=32" >> $OUT/s1_convt16_layers.txt
GG_CONVT16=64 GG_CONVT16_TW=32 ITERS=20 timeout 300 python scripts/conv_bench.py 'upconv ' >> $OUT/s1_convt16_layers.txt 2>&1
grep -v relerr $OUT/s1_convt16_layers.txt | tail -n 5
grep "upconv" $OUT/s1_convt16_layers.txt | grep -v dgrad
# the step, same box: round-4 tiles / new tile where a layer gives every CU two blocks / new tile on every transposed launch
for mode in 0 1 64 0 1; do
  GG_CONVT16=$mode timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/s1_bench_mode${mode}_$RANDOM.json 2> $OUT/s1_bench_err.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05/s1_bench_mode*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        ks = {k['kernel']: (k['ms_per_step'], k['rate']) for k in d['roofline'].get('kernels', [])}
        print(f, d['ms_per_step'], d['value'], {k: v for k, v in ks.items() if 'convT' in k})
    except Exception as e:
        print(f, 'unreadable', e)
PY
