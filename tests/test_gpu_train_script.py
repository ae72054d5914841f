"""The reference's train.py ITSELF, unmodified, as a process under the launcher:

    python -m gangealing_amd.launch [--modules] <reference>/train.py --exp-name ... --ckpt <synthetic> --load_G_only ...

north_star: "... so train.py is a drop-in".  Earlier rounds ran the loop body re-typed in a test; this runs the
script: argument parsing (utils/base_argparse.py), results directory + opt.txt, model construction, the checkpoint load
(train.py:216-228: `g_ema` only, then PCA over 1000 mapped latents for the DirectionInterpolator), optimizers and
DecayingCosineAnnealingWarmRestarts, psi annealing, the training loop (train.py:89-134), learning-rate restarts, EMA,
checkpoints at the zero-learning-rate iterations (save_state_dict, train.py:22-28), TensorBoard scalars and the training
visuals (utils/vis_tools/training_vis.py: generator + EMA STN under inference_mode).  What the offline box lacks is
stood in for by the launcher (gangealing_amd/launch.py: stub_missing): torchvision's VGG16 layer list, make_grid and a
SummaryWriter that appends scalars to scalars.jsonl.  The generator checkpoint and the SimCLR VGG16 weights are
synthetic files written here in the reference's layouts (no network).

World size 2 (both ranks on the one GPU, gloo): the reference's own DistributedDataParallel wrap (train.py:256-259)
around the STN built from HIP operators - its bucketed gradient all-reduce hooks run against our autograd functions."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu

ARGS = ['--exp-name', 'synthetic', '--load_G_only', '--debug', '--gen_size', '64', '--real_size', '64', '--flow_size', '64',
        '--batch', '2', '--iter', '12', '--anneal_psi', '4', '--period', '2', '--ckpt_every', '5', '--vis_every', '6',
        '--log_every', '2', '--n_sample', '4', '--n_mean', '8', '--vis_batch_size', '4', '--inject', '3', '--ndirs', '2',
        '--stn_lr', '0.001', '--ll_lr', '0.01', '--seed', '3']


def reference_root():
    from oracle import pyref
    root = pyref.find_root()
    if root is None:
        pytest.skip('reference Python not staged (run `make -C oracle` where /root/reference exists)')
    return root


def write_synthetic_inputs(workdir):
    """g_ema checkpoint in the reference's layout (train.py:216-217 loads ckpt['g_ema'] strictly) and the VGG16
    `features` state_dict that lpips_backbones.py:103-105 loads strictly from pretrained/simclr_vgg_phase150.pt."""
    from gangealing_amd.stylegan2 import Generator
    from gangealing_amd._standins import vgg16
    torch.manual_seed(11)
    g = Generator(64, 512, 8, channel_multiplier=2)
    ckpt = os.path.join(workdir, 'g_synthetic.pt')
    torch.save({'g_ema': g.state_dict()}, ckpt)
    feats = vgg16().features
    for m in feats:
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight, nonlinearity='relu')
            torch.nn.init.normal_(m.bias, std=0.05)
    os.makedirs(os.path.join(workdir, 'pretrained'), exist_ok=True)
    torch.save(feats.state_dict(), os.path.join(workdir, 'pretrained', 'simclr_vgg_phase150.pt'))
    return ckpt


def check_run(workdir, res, world=1):
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert 'Only G_EMA has been loaded from checkpoint' in res.stdout
    out = os.path.join(workdir, 'results', 'synthetic')
    opt = json.load(open(os.path.join(out, 'opt.txt')))
    assert opt['batch'] == 2 and opt['distributed'] == (world > 1)
    # scalars the script logged (train.py:145-150): finite, and the schedule visible in them
    rows = [json.loads(ln) for ln in open(os.path.join(out, 'scalars.jsonl'))]
    by_tag = {}
    for r in rows:
        by_tag.setdefault(r['tag'], []).append((r['step'], r['value']))
    for tag in ('Loss/Reconstruction', 'Loss/TotalVariation', 'Loss/FlowIdentity', 'Progress/psi',
                'Progress/STN_LearningRate'):
        assert tag in by_tag and len(by_tag[tag]) >= 5, tag
        assert all(np.isfinite(v) for _, v in by_tag[tag]), (tag, by_tag[tag])
    assert all(v > 0 for _, v in by_tag['Loss/Reconstruction'])
    psi = dict(by_tag['Progress/psi'])
    assert psi[2] > psi[4] == 0.0                                  # annealed to zero over --anneal_psi iterations
    # checkpoints: every --ckpt_every and at the zero-learning-rate iterations (utils/annealing.py: lr_cycle_iters)
    ckpts = sorted(os.listdir(os.path.join(out, 'checkpoints')))
    assert '0000005.pt' in ckpts and '0000010.pt' in ckpts, ckpts
    # training visuals written by the script's own code (PNG grids through the make_grid stand-in)
    pngs = [f for f in os.listdir(out) if f.endswith('.png')]
    assert any(f.startswith('transformed_sample_0000000') for f in pngs) and any('0000006' in f for f in pngs), pngs
    return os.path.join(out, 'checkpoints', ckpts[-1])


def check_checkpoint(path, cuda):
    """The file train.py wrote (train.py:22-28) loads into this package's trainer (GangealingTrainer.load_state_dict =
    train.py:216-228) and training continues from it."""
    from gangealing_amd.train_step import GangealingTrainer
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    assert set(ckpt) == {'g_ema', 't', 't_ema', 't_optim', 't_sched', 'll', 'll_optim', 'll_sched', 'args'}
    tr = GangealingTrainer(cuda, gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3, ndirs=2,
                           seed=5)
    assert tr.load_state_dict(ckpt) is True
    sd = tr.stn.state_dict()
    for k, v in ckpt['t'].items():
        assert torch.equal(sd[k].cpu(), v), k
    assert int(ckpt['t_optim']['state'][0]['step']) == tr.stn_arena.step_count > 0
    moved = max(float((a - b).abs().max()) for a, b in zip(ckpt['t'].values(), ckpt['t_ema'].values()))
    assert moved > 0                                               # the EMA copy lags the trained STN
    p = tr.step(psi=0.0)
    assert np.isfinite(float(p['p']))


@pytest.mark.parametrize('route', ['literal', 'modules'])
def test_train_py_runs_under_the_launcher(route, cuda, tmp_path):
    root = reference_root()
    workdir = str(tmp_path)
    ckpt = write_synthetic_inputs(workdir)
    cmd = [sys.executable, '-m', 'gangealing_amd.launch'] + (['--modules'] if route == 'modules' else []) + \
          [os.path.join(root, 'train.py'), '--ckpt', ckpt, '--results', os.path.join(workdir, 'results')] + ARGS
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get('PYTHONPATH', ''), GANGEALING_SYNTHETIC='1',
               GANGEALING_CONV_PRECISION='fp16x3')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    res = subprocess.run(cmd, env=env, cwd=workdir, capture_output=True, text=True, timeout=900)
    last = check_run(workdir, res)
    check_checkpoint(last, cuda)


def test_train_py_two_ranks_through_the_reference_ddp_wrap(cuda, tmp_path):
    root = reference_root()
    workdir = str(tmp_path)
    ckpt = write_synthetic_inputs(workdir)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(REPO, 'tests', 'run_train_script_rank.py'),
           os.path.join(root, 'train.py'), '--ckpt', ckpt, '--results', os.path.join(workdir, 'results')] + ARGS
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get('PYTHONPATH', ''), GANGEALING_SYNTHETIC='1',
               GANGEALING_CONV_PRECISION='fp16x3', HSA_ENABLE_IPC_MODE_LEGACY='0')
    res = subprocess.run(cmd, env=env, cwd=workdir, capture_output=True, text=True, timeout=900)
    last = check_run(workdir, res, world=2)
    ckpt2 = torch.load(last, map_location='cpu', weights_only=False)
    assert ckpt2['args'].distributed is True and not any(k.startswith('module.') for k in ckpt2['t'])
