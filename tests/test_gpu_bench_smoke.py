"""bench.py end to end under the driver's own launch line, at world size 2: `python -m torch.distributed.run --nnodes=1
--nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`.  A test box has ONE GPU, so both
ranks share cuda:0 and the process group is gloo (RCCL refuses two ranks on one device) - GANGEALING_SHARE_DEVICE /
GANGEALING_DIST_BACKEND, developer switches the driver never sets; everything else (rendezvous, barriers, the
max-over-ranks timing, the asynchronous all-reduce of the gradient arena with the deferred optimizer update, the
JSON line with its `distributed` block) is the code path of an 8-GPU run."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_one_gpu(cuda):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, GANGEALING_SHARE_DEVICE='1', GANGEALING_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '3',
           '--warmup', '1', '--workload', 'c1', '--no-extras', '--no-cpu-baseline']
    res = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]                      # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['scaling'] == 'weak' and out['value'] > 0
    assert out['config']['global_batch'] == 2 * 4 and out['config']['parallelism'] == 'dp2'
    d = out['distributed']
    assert d['world_size'] == 2 and d['ranks_counted_by_all_reduce'] == 2 and d['backend'] == 'gloo'
    assert d['pipelined_update'] is True and d['allreduce_bytes'] > 1e6
    assert 0 < d['ms_per_step_rank_min'] <= d['ms_per_step_rank_max']
    assert d['allreduce_exposed_ms_per_step_max_rank'] >= 0
    # whole-job throughput = images of BOTH ranks over the max-over-ranks time
    assert abs(out['value'] - 2 * 4 * 3 / (out['ms_per_step'] * 3e-3)) / out['value'] < 1e-3


def test_bench_allreduce_only_two_ranks(cuda):
    """`bench.py --gpus 2 --allreduce-only`: the step's one data-path collective alone (the STN gradient arena, 172 MB at C2),
    so that a poor scaling curve can be split into "RCCL" and "the step"."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, GANGEALING_SHARE_DEVICE='1', GANGEALING_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1',
           '--warmup', '1', '--allreduce-only']
    res = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['backend'] == 'gloo' and out['iters'] == 5
    assert out['bytes'] > 160e6 and out['ms_mean_max_over_ranks'] > 0 and out['algbw_GBps'] > 0
    assert len(out['per_rank_ms_mean_min_max_wall']) == 2
    assert abs(out['busbw_GBps'] - out['algbw_GBps']) < 1e-6 * max(out['algbw_GBps'], 1) + 0.02       # 2 (w - 1) / w = 1
