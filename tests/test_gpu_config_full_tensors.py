"""EVERY element of the benchmark configuration's activations against the reference - not the fixtures' strided sample.

tests/golden/cfg_c2.npz stores sample 0 in full, a 1-in-N strided subsample of the other fifteen and per-sample sums
(oracle/config_cases.pack_batch): a fault confined to a few pixels of samples 1 - 15 - a tile edge, one image row - could
sit between the strides, and the sums (bounded at 1e-4 sum|x|) would not notice it.  Storing the full batch-16 tensors
would add ~25 MB to a repository that already carries too many fixtures, so this test produces them where they are
needed: the reference's own Python, staged under oracle/_ref/pyref (it travels with gpurun; oracle/Makefile), evaluates
the forward pass of config C2 at batch 16 on this box's HOST cores, in a child process (the reference's modules and the
drop-in tests both bind `models.*`), through the same driver that wrote the fixtures (oracle/config_cases.run_config);
the HIP path runs the same step in this process; all five activation tensors are compared element by element at
north_star's 1e-4, in the benched arithmetic and the exact-fp32 one.  The child's sample-0 tensors must equal the committed
fixture's to rounding, which ties the live run to the vectors the rest of the suite uses.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden, record_parity

pytestmark = pytest.mark.gpu

KEYS = ('unaligned', 'target', 'pred', 'stn_delta', 'delta_flow')

CHILD = r'''
import sys
sys.path.insert(0, %r)
sys.dont_write_bytecode = True
import numpy as np
import torch
torch.set_num_threads(%d)
from oracle import pyref, config_cases as cc
api = pyref.cpu_api()
assert api is not None
res = cc.run_config(api, 'c2', 'cpu', backward=False)
np.savez(%r, **{k: res[k].numpy() for k in %r}, ploss=res['ploss'].numpy(), tv=res['tv'].numpy(),
         identity=res['identity'].numpy())
print('reference-forward-done')
'''


@pytest.fixture(scope='module')
def reference_c2(tmp_path_factory):
    from oracle import pyref
    if pyref.find_root() is None:
        pytest.skip('reference Python not staged (run `make -C oracle` where /root/reference exists)')
    out = str(tmp_path_factory.mktemp('c2full') / 'ref.npz')
    threads = min(os.cpu_count() or 8, 32)
    env = dict(os.environ, CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
    res = subprocess.run([sys.executable, '-c', CHILD % (REPO, threads, out, KEYS)], env=env, capture_output=True,
                         text=True, timeout=1500)
    assert res.returncode == 0 and 'reference-forward-done' in res.stdout, res.stderr[-3000:]
    ref = dict(np.load(out))
    # the live run is the run behind the committed fixture (another host's fp32 kernels: equal to rounding, not bitwise)
    (fix,) = load_golden('cfg_c2')
    for key in KEYS:
        a, b = ref[key][0], fix[f'{key}_first']
        assert a.shape == b.shape and float(np.abs(a - b).max()) <= 2e-5 * max(1.0, float(np.abs(b).max())), key
    return ref


@pytest.mark.parametrize('mode', ['fp16x3', 'fp32'])
def test_config_c2_every_element_of_every_sample(mode, reference_c2, cuda):
    from gangealing_amd.op import conv_mfma
    from oracle import config_cases as cc
    from test_gpu_configs import our_api
    old = conv_mfma.PRECISION
    conv_mfma.set_precision(mode)
    try:
        res = cc.run_config(our_api(), 'c2', cuda, backward=False)
    finally:
        conv_mfma.set_precision(old)
    for key in KEYS:
        got, ref = res[key].cpu().numpy(), reference_c2[key]
        assert got.shape == ref.shape and got.shape[0] == 16, (key, got.shape)
        err = record_parity('cfg_c2_full', mode, key, got, ref, extra=dict(elements=int(ref.size)))
        scale = max(1.0, float(np.abs(ref).max()))
        assert err <= 1e-4 * scale, (key, err, scale)
        # ... and sample by sample, so that the report shows WHERE the worst one sits
        per_sample = np.abs(got - ref).reshape(16, -1).max(1)
        assert (per_sample <= 1e-4 * scale).all(), (key, per_sample)
    for key in ('ploss', 'tv', 'identity'):
        ref = float(reference_c2[key])
        assert abs(float(res[key]) - ref) <= 1e-4 * abs(ref) + 1e-9, (key, float(res[key]), ref)
