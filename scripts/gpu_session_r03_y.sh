#!/bin/bash
# Round-3 session Y: non-temporal epilogue stores of the 8-wave tiles for outputs larger than the Infinity Cache
# (GG_NT_STORE=0 never | unset: outputs > 256 MB | 1 always), per layer and on the whole step
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03y
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
timeout 600 python -m pytest tests/test_gpu_c2_layer_ops.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -3 > $O/pytest.txt
for m in 0 auto 1; do
  if [ $m = auto ]; then unset GG_NT_STORE; else export GG_NT_STORE=$m; fi
  GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 python scripts/conv_bench.py "G " > $O/layers_$m.txt 2>&1
  for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_${m}_$i.json 2>/dev/null; done
done
cat $O/pytest.txt
paste <(grep "G conv\|G upconv" $O/layers_0.txt | grep -v dgrad | cut -c1-100) <(grep "G conv\|G upconv" $O/layers_auto.txt | grep -v dgrad | cut -c66-100) <(grep "G conv\|G upconv" $O/layers_1.txt | grep -v dgrad | cut -c66-100)
for f in $O/bench_*.json; do echo -n "$f "; head -c 175 $f | tail -c 60; echo; done
