#!/usr/bin/env python
"""Headline benchmark: GANgealing train-step images/sec (BASELINE.json metric) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = loss forward (G x2, STN, perceptual loss, regularisers), backward (incl. the G pass-2
backward), gradient all-reduce over the flat arena, Adam x2 and the EMA update (train.py:106-134),
on random-latent batches with inputs already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL's buffer registration fails with the legacy mode:
# hipIpcGetMemHandle: invalid argument); the environment normally carries it already
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: LSUN Cats 256x256, similarity+flow STN (flow_size=128), batch 16 per GPU
    'c2': dict(gen_size=256, flow_size=128, batch=16, transform=('similarity', 'flow'), num_heads=1, flips=False,
               inject=5, ndirs=1, padding_mode='reflection', sample_from_full_res=False),
    # configs[3]: CelebA-HQ 512x512 flags (scripts/training/celeba.sh:4-6): BilinearDownsample(4) to the 128^2 STN,
    # border padding, full-resolution sampling source, inject 6, ndirs 512 (high-res upfirdn2d stress); per-GPU batch 4
    'c4': dict(gen_size=512, flow_size=128, batch=4, transform=('similarity', 'flow'), num_heads=1, flips=False,
               inject=6, ndirs=512, padding_mode='border', sample_from_full_res=True, tv_weight=2500.0),
    # configs[4]: LSUN Cars clustering (scripts/training/lsun_cars.sh:4-7): K = 4 heads with flips (the STN and the
    # perceptual loss see 8x the batch), full-resolution sampling, inject 6, ndirs 5; per-GPU batch 4
    'c5': dict(gen_size=256, flow_size=128, batch=4, transform=('similarity', 'flow'), num_heads=4, flips=True,
               inject=6, ndirs=5, padding_mode='reflection', sample_from_full_res=True, tv_weight=2500.0),
    # configs[0]: plumbing (64x64, similarity only)
    'c1': dict(gen_size=64, flow_size=64, batch=4, transform=('similarity',), num_heads=1, flips=False, inject=5,
               ndirs=1, padding_mode='reflection', sample_from_full_res=False, tv_weight=0.0, flow_identity_weight=0.0),
}

METRIC = {'c2': 'train-step images/sec, LSUN-Cats 256^2 STN+StyleGAN2',
          'c1': 'train-step images/sec, LSUN-Cats 64^2 similarity STN+StyleGAN2 (plumbing configuration)',
          'c4': 'train-step images/sec, CelebA-HQ 512^2 STN+StyleGAN2 (BASELINE configs[3] flags)',
          'c5': 'train-step images/sec, LSUN-Cars 256^2 K=4 clustering STN+StyleGAN2 (BASELINE configs[4] flags)'}

# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks
MFMA_PEAK_TFLOPS = {'fp32': 157.3, 'bf16': 2500.0, 'bf16x3': 2500.0, 'bf16x6': 2500.0, 'fp16x3': 2500.0}
# descriptions only (roofline.kernel_detail); roofline.kernel is the name the dispatcher reports at run time
KERNEL_NAME = {
    'fp32': 'conv_igemm_kernel<3,0,2,2,2,2,*> (3x3 correlation, 128co x 128pix tile, v_mfma_f32_32x32x2_f32)',
    'bf16x3': 'conv3x3_patch_kernel<2, true, 256, 2, 0, 1> (3x3 stride-1 modulated conv + fused noise/bias/lrelu epilogue, '
              '128co x 256pix tile, input patch staged once per 32-channel chunk, double-buffered weight slab with a '
              'software-pipelined tap loop, v_mfma_f32_32x32x16_bf16, 2 bf16 limbs per fp32 operand = 3 MFMA products per '
              'algorithmic product)',
    'fp16x3': 'conv3x3_patch_kernel<2, true, 256, 2, 0, 1, true> (the bf16x3 tile with binary16 limbs: '
              'v_mfma_f32_32x32x16_f16, 2 limbs per fp32 operand = 3 MFMA products per algorithmic product, weights '
              'pre-scaled by 2^8 in the pack, a power-of-two block exponent per tile; data gradients the same, weight '
              'gradients on bf16 limbs)',
    'bf16': 'conv3x3_patch_kernel<1, true, 256, 2, 0, 3> (same tile as bf16x3, one bf16 limb per operand = one MFMA '
            'product per algorithmic product, fp32 accumulate; three tap slabs staged per barrier interval)',
    'bf16x6': 'conv3x3_patch_kernel<3, true, 128, 2, 0, 1> (same layers, 128co x 128pix tile, 3 bf16 limbs per fp32 operand '
              '= 6 MFMA products per algorithmic product)',
}
MFMA_PRODUCTS = {'fp32': 1, 'bf16': 1, 'bf16x3': 3, 'bf16x6': 6, 'fp16x3': 3}


def kernel_source_hash():
    """sha256 (16 hex digits) over the convolution kernel source and the shared headers: identifies the code a
    profile was taken with."""
    import hashlib
    h = hashlib.sha256()
    for rel in ('gangealing_amd/csrc/conv_mfma.hip', 'gangealing_amd/csrc/conv_common.h', 'gangealing_amd/csrc/conv_s2_patch.hip',
                'gangealing_amd/csrc/conv_s2_wgrad.hip', 'gangealing_amd/csrc/conv_t_c16.hip', 'gangealing_amd/csrc/gg_common.h', 'include/gangealing_hip.h'):
        with open(os.path.join(REPO, rel), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


PMC_RECORDS = ('r06_pmc_traffic.json', 'r06_pmc_traffic_c4.json', 'r06_pmc_traffic_c5.json', 'r05_pmc_traffic.json',
               'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json')


def pmc_traffic(precision, workload, batch):
    """HBM bytes per launch of the dominant kernel, measured offline with rocprofv3 --pmc on this same command
    (scripts/measure_r04.sh) and committed under profiles/.  None for configurations that were not profiled AND
    whenever the recorded kernel-source hash differs from the tree's (a profile of another version of the kernel says
    nothing about this one)."""
    here = os.path.dirname(os.path.abspath(__file__))
    for name in PMC_RECORDS:
        try:
            with open(os.path.join(here, 'profiles', name)) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if (rec.get('precision'), rec.get('workload'), rec.get('batch')) == (precision, workload, batch):
            if rec.get('kernel_source_sha16') != kernel_source_hash():
                return None
            return rec.get('hbm_bytes_per_launch')
    return None


def pmc_traffic_source(precision, workload, batch):
    """Where roofline.traffic comes from: it is RECORDED (a rocprofv3 --pmc session cannot run inside this process),
    and replayed only while the kernel sources hash to what was profiled."""
    for name in PMC_RECORDS:
        try:
            with open(os.path.join(REPO, 'profiles', name)) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if (rec.get('precision'), rec.get('workload'), rec.get('batch')) == (precision, workload, batch):
            if rec.get('kernel_source_sha16') != kernel_source_hash():
                return f'none: profiles/{name} was recorded for other kernel sources ({rec.get("kernel_source_sha16")})'
            return (f'recorded: profiles/{name} ({rec.get("command", "rocprofv3 --pmc")}; {rec.get("commit", "")}), replayed '
                    f'because the kernel-source hash {rec.get("kernel_source_sha16")} equals this tree\'s')
    return 'none: this configuration was not profiled'


DTYPE = {'fp32': 'f32', 'bf16': 'bf16 (convolution operands rounded to bf16, fp32 accumulate, fp32 activations)', 'bf16x3': 'bf16x3 (fp32 operands split into 2 bf16 limbs, fp32 accumulate)',
         'bf16x6': 'bf16x6 (fp32 operands split into 3 bf16 limbs, fp32 accumulate)',
         'fp16x3': 'fp16x3 (fp32 operands split into 2 sixteen-bit limbs, 3 MFMA products, fp32 accumulate: binary16 limbs with '
                   'a per-tile block exponent on the forward and data-gradient convolutions, bf16 limbs on the weight gradients)'}


def _reference_step_fn(wl):
    """The REFERENCE's own training-loss step on its pure-PyTorch CPU op fallback (upfirdn2d_native, CPU
    fused_leaky_relu, F.conv2d / grid_sample): its modules are imported through the stub loader of
    oracle/make_golden.py from the checkout (authoring container) or from oracle/_ref/pyref, the copy `make -C oracle`
    stages so that it travels to the GPU box (oracle/pyref.py).  None when neither exists."""
    try:
        from oracle import pyref
        api = pyref.cpu_api()
    except Exception:                          # noqa: BLE001 - any import problem: fall back to the port
        return None
    if api is None:
        return None
    gen = api.Generator(wl['gen_size'], 512, 8).eval().requires_grad_(False)
    stn = api.get_stn(list(wl['transform']), flow_size=wl['flow_size'], supersize=wl['gen_size'],
                      channel_multiplier=0.5, num_heads=1)
    ema = api.get_stn(list(wl['transform']), flow_size=wl['flow_size'], supersize=wl['gen_size'],
                      channel_multiplier=0.5, num_heads=1)
    ll = api.DirectionInterpolator(None, wl['ndirs'], wl['inject'], gen.n_latent)
    t_optim = torch.optim.Adam(stn.parameters(), lr=1e-3)
    ll_optim = torch.optim.Adam(ll.parameters(), lr=1e-2)
    mse = lambda x, y: ((x - y) ** 2).mean(dim=(1, 2, 3))         # torchvision (LPIPS trunk) is not installed

    def step():
        loss, _ = api.gangealing_loss(gen, stn, ll, mse, torch.nn.Sequential(), 0.5, wl['batch'], 512, False, 'cpu',
                                      padding_mode=wl['padding_mode'])
        stn.zero_grad()
        ll.zero_grad()
        loss.backward()
        t_optim.step()
        ll_optim.step()
        api.accumulate(ema, stn, 0.5 ** (32 / 10000))
    step.source = api.root
    return step


def _port_step_fn(wl):
    """oracle/torch_ref.py: the torch-CPU restatement of the same step (what travels to the GPU box)."""
    from oracle import torch_ref as R
    from oracle.det_weights import det_state_dict
    from gangealing_amd.stylegan2 import Generator
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    g_sd = det_state_dict(Generator(wl['gen_size'], 512, 8))
    stn = get_stn(list(wl['transform']), flow_size=wl['flow_size'], supersize=wl['gen_size'], channel_multiplier=0.5)
    stn_sd = {k: v.clone().requires_grad_(True) for k, v in
              det_state_dict(stn, (('warp_head.linear', 0.02),)).items()}
    ll_sd = dict(directions=torch.randn(1, 512), lat_mean=torch.randn(1, 512),
                 coefficients=torch.zeros(1, 1, requires_grad=True))
    params = list(stn_sd.values()) + [ll_sd['coefficients']]
    opt = torch.optim.Adam(params, lr=1e-3)
    ema = [p.detach().clone() for p in stn_sd.values()]
    decay = 0.5 ** (32 / 10000)

    def step():
        z = torch.randn(wl['batch'], 512)
        total, _ = R.train_loss(g_sd, stn_sd, ll_sd, z, wl['gen_size'], wl['flow_size'], 0.5, wl['inject'],
                                wl['padding_mode'], wl['transform'], R.mse_loss_fn, 0.0, 0.0)
        opt.zero_grad()
        total.backward()
        opt.step()
        with torch.no_grad():
            for e, p in zip(ema, stn_sd.values()):
                e.mul_(decay).add_(p, alpha=1 - decay)
    return step


def cpu_baseline(budget_s=20.0):
    """Config C1 (gen 64, similarity-only STN at 64, batch 4: BASELINE.json configs[0]) as full train steps - loss
    forward, backward, Adam x2, EMA - on this box's host cores, MSE stand-in for the VGG loss.  kind "reference" when
    the reference checkout is importable (its own code runs), else "port" (oracle/torch_ref.py).  The thread count is
    chosen by a 3-point sweep (one step each) and reported; the sample is whole steps until ~budget_s seconds."""
    wl = WORKLOADS['c1']
    step = _reference_step_fn(wl)
    kind = 'reference' if step is not None else 'port'
    if step is None:
        step = _port_step_fn(wl)
    cores = os.cpu_count() or 1
    sweep = {}
    for threads in sorted({min(cores, t) for t in (8, 16, 32)}):
        torch.set_num_threads(threads)
        step()                                     # warm (allocator, thread pool)
        t0 = time.perf_counter()
        step()
        sweep[threads] = time.perf_counter() - t0
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    steps, t0 = 0, time.perf_counter()
    while True:
        step()
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or steps >= 50:
            break
    src = (f'the reference\'s own modules on its pure-PyTorch CPU op fallback (imported from {os.path.relpath(step.source, REPO) if step.source.startswith(REPO) else step.source})'
           if kind == 'reference' else 'oracle/torch_ref.py (no reference Python on this box)')
    return dict(value=round(steps * wl['batch'] / dt, 3), unit='images/sec', cores=threads, kind=kind,
                sample=f'{steps} full train steps of config C1 (gen 64, similarity STN@64, batch {wl["batch"]}, '
                       f'MSE stand-in loss) in {dt:.1f} s via {src}; {threads} of {cores} host cores, chosen by a '
                       f'one-step sweep: ' + ', '.join(f'{t} threads {v:.2f} s' for t, v in sorted(sweep.items())))


def cpu_baseline_child(budget_s):
    """cpu_baseline() in a child interpreter: the reference's modules on their CPU fallback and the same modules on the
    HIP operators (extras.dropin_route) both bind `models.*` in sys.modules, so one process holds only one of them."""
    import subprocess
    env = dict(os.environ, CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
    out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--cpu-budget', str(budget_s)],
                         capture_output=True, text=True, env=env, timeout=max(600.0, 30 * budget_s))
    for line in reversed(out.stdout.strip().splitlines()):
        if line.startswith('{'):
            return json.loads(line)
    raise RuntimeError('cpu baseline child failed: ' + out.stderr[-400:])


SYNTHETIC_LR = 1e-4      # the work per step does not depend on the learning rate; with randomly initialised G / VGG the
                         # reference's 1e-3 sends the random STN to extreme zooms within a few iterations (gradient spikes of
                         # 1e4 .. 1e10 in the similarity head), 1e-4 keeps it near its perturbed initial warp


def measure(device, wl, precision, graph, steps, warmup, world, gdist, profile=True, modconv='shared'):
    """Warm-up, then time exactly `steps` iterations between barrier + synchronize pairs.  -> dict.
    modconv='grouped': the generator's modulated convolutions in the reference's own formulation (materialised
    per-sample weights + grouped convolution through op.conv2d_gradfix: the literal drop-in route)."""
    from gangealing_amd.stylegan2 import networks
    with networks.modconv_form(modconv):
        return _measure(device, wl, precision, graph, steps, warmup, world, gdist, profile)


def _measure(device, wl, precision, graph, steps, warmup, world, gdist, profile):
    from gangealing_amd.op import conv_mfma
    workload_is_c2 = (wl['gen_size'], wl['flow_size'], wl['num_heads'], wl['flips']) == (256, 128, 1, False)
    from gangealing_amd.train_step import GangealingTrainer
    conv_mfma.set_precision(precision)
    trainer = GangealingTrainer(device, perturb_heads=0.02, seed=0, use_graph=graph,
                                stn_lr=SYNTHETIC_LR, ll_lr=SYNTHETIC_LR, allow_random_loss=True, **wl)

    def barrier():
        torch.cuda.synchronize()
        gdist.synchronize()
        torch.cuda.synchronize()

    graphed = trainer.use_graph
    for _ in range(max(warmup, trainer._graph_warmup + 1 if graphed else 0)):
        trainer.step(psi=0.5)
    trainer.flush()
    barrier()
    # Which kernel dominates this workload?  Two un-timed steps with HIP events around EVERY convolution / FIR launch,
    # keyed by the kernel instantiation the library reports (gg_last_conv_kernel).  Config C2 at its benchmark batch
    # keeps the launch predicate validated against rocprofv3 (conv_mfma.conv_forward); every other workload times, in
    # its timed region, the kernel the survey found on top.
    survey, dominant, step_classes = None, None, None
    if profile and not graphed and world == 1:
        sv = conv_mfma.LaunchProfiler(every=True)
        conv_mfma.PROFILER = sv
        for _ in range(2):
            trainer.step(psi=0.5)
        trainer.flush()
        conv_mfma.PROFILER = None
        table = sv.by_kernel()
        # whole-step classes (roofline.step): every convolution launch is MFMA-class work with its algorithmic FLOPs,
        # every FIR / blur launch HBM-class work with its algorithmic bytes; everything else of the step (streaming
        # activations, samplers, optimizer, ATen glue) is what remains of ms_per_step
        step_classes = dict(
            mfma_ms=sum(v['ms'] for v in table.values() if v['unit'] == 'flop') / 2,
            mfma_flop=sum(v['work'] for v in table.values() if v['unit'] == 'flop') / 2,
            mfma_launches=sum(v['launches'] for v in table.values() if v['unit'] == 'flop') / 2,
            hbm_ms=sum(v['ms'] for v in table.values() if v['unit'] == 'byte') / 2,
            hbm_bytes=sum(v['work'] for v in table.values() if v['unit'] == 'byte') / 2,
            hbm_launches=sum(v['launches'] for v in table.values() if v['unit'] == 'byte') / 2)
        survey = [dict(kernel=k, launches_per_step=v['launches'] / 2, ms_per_step=round(v['ms'] / 2, 4),
                       rate=round(v['work'] / (v['ms'] * 1e-3) / 1e12, 3) if v['ms'] > 0 else 0.0,
                       unit='TFLOP/s' if v['unit'] == 'flop' else 'TB/s')
                  for k, v in sorted(table.items(), key=lambda kv: -kv[1]['ms'])[:int(os.environ.get('GG_BENCH_SURVEY_ROWS', 8))]]
        if survey and not (workload_is_c2 and wl['batch'] >= 8):
            dominant = survey[0]['kernel']
        barrier()
    prof = conv_mfma.LaunchProfiler(only=dominant, names=sv.names) if dominant else conv_mfma.LaunchProfiler()
    if profile and not graphed:
        conv_mfma.PROFILER = prof        # HIP events around the dominant kernel's launches inside the timed region
    if world > 1:
        trainer.comm_events = []
    from gangealing_amd import _lib as _gglib
    calls0 = _gglib.CALLS
    t0 = time.perf_counter()
    for _ in range(steps):
        parts = trainer.step(psi=0.5)
    trainer.flush()                      # a deferred (pipelined) optimizer step belongs to the timed region
    lib_calls = (_gglib.CALLS - calls0) / max(steps, 1)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0       # this rank's own time, before waiting for the others
    barrier()
    elapsed = time.perf_counter() - t0
    conv_mfma.PROFILER = None
    dist_info = None
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # what the collective library itself saw: an all-reduce of ones counts the ranks that took part
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        per_rank = [torch.zeros(1, device=device, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([own], device=device, dtype=torch.float64))
        exposed = [a.elapsed_time(b) for a, b in trainer.comm_events]
        ex = torch.tensor([sum(exposed) / max(len(exposed), 1)], device=device, dtype=torch.float64)
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        ms = [1e3 * float(v.item()) / steps for v in per_rank]
        dist_info = dict(backend=dist.get_backend(), world_size=dist.get_world_size(),
                         ranks_counted_by_all_reduce=int(round(float(ones.item()))),
                         ms_per_step_rank_min=round(min(ms), 3), ms_per_step_rank_max=round(max(ms), 3),
                         allreduce_exposed_ms_per_step_max_rank=round(float(ex.item()), 4),
                         allreduce_bytes=int(trainer.stn_arena.numel * 4), pipelined_update=bool(trainer.pipeline_update),
                         ms_per_step_per_rank=[round(v, 3) for v in ms],
                         collective_env={k: v for k, v in sorted(os.environ.items())
                                         if k.startswith(('NCCL_', 'RCCL_', 'HSA_', 'TORCH_NCCL_')) or k == 'HIP_VISIBLE_DEVICES'},
                         devices=sorted({torch.cuda.current_device()}))
    loss = float(parts['p'])
    assert loss == loss and abs(loss) != float('inf'), 'non-finite loss'
    reported = sorted({r[3] for r in prof.records}) if (profile and not graphed) else None
    res = dict(elapsed=elapsed, steps=steps, loss=loss, graphed=graphed, prof=prof.summary() if (profile and not graphed) else None,
               reported=reported,
               images=world * wl['batch'] * steps, dist=dist_info, survey=survey, dominant=dominant,
               step_classes=step_classes, lib_calls=lib_calls,
               dominant_unit=(prof.records[0][4] if (dominant and prof.records) else 'flop'))
    del trainer
    torch.cuda.empty_cache()
    return res


def measure_reference_dropin(device, wl, precision, steps, warmup, modules=False):
    """The literal drop-in route: the REFERENCE's own modules (networks.py with its per-sample-weight / groups = N
    modulated convolutions, spatial_transformer.py, warping_heads.py, latent_learner.py, loss.py, lpips.py),
    imported unmodified through gangealing_amd.launch.inject (oracle/pyref.hip_api), run the iteration of
    train.py:106-134 on the HIP operators with torch.optim.Adam and models.accumulate - what a user of
    `python -m gangealing_amd.launch train.py` gets.  None when no reference Python is on this box."""
    from oracle import pyref                        # checker-side loader; the timed modules are the reference's
    api = pyref.hip_api(modules=modules)        # modules=True: `python -m gangealing_amd.launch --modules train.py`
    if api is None:
        return None
    from gangealing_amd.op import conv_mfma
    conv_mfma.set_precision(precision)
    torch.manual_seed(0)
    gen = api.Generator(wl['gen_size'], 512, 8, channel_multiplier=2).to(device).eval().requires_grad_(False)
    kw = dict(flow_size=wl['flow_size'], supersize=wl['gen_size'], channel_multiplier=0.5, num_heads=wl['num_heads'])
    stn = api.get_stn(list(wl['transform']), **kw).to(device)
    with torch.no_grad():                           # as GangealingTrainer(perturb_heads=0.02): a non-identity warp
        for name, p in stn.named_parameters():
            if 'warp_head.linear' in name or 'flow_out.2' in name:
                p.normal_(0.0, 0.02)
    ema = api.get_stn(list(wl['transform']), **kw).to(device)
    api.accumulate(ema, stn, 0)
    ll = api.DirectionInterpolator(None, wl['ndirs'], wl['inject'], gen.n_latent, wl['num_heads']).to(device)
    net = api.LPIPS(net='vgg', lpips=False, pnet_rand=True, pretrained=False, verbose=False).to(device).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.Conv2d) and m.bias is not None:
                m.bias.fill_(0.1)                   # as losses.py: no all-zero feature vectors under a random trunk
    loss_fn = lambda x, y: net(x, y) / 18.0        # lpips.py:17
    factor = wl['gen_size'] // wl['flow_size']
    resize = api.BilinearDownsample(factor, 3).to(device) if factor > 1 else torch.nn.Sequential()
    t_optim = torch.optim.Adam(stn.parameters(), lr=SYNTHETIC_LR, betas=(0.9, 0.999), eps=1e-8)
    ll_optim = torch.optim.Adam(ll.parameters(), lr=SYNTHETIC_LR, betas=(0.9, 0.999), eps=1e-8)
    clustering = wl['num_heads'] > 1 or wl['flips']
    tv_w, id_w = wl.get('tv_weight', 1000.0), wl.get('flow_identity_weight', 1.0)
    common = dict(sample_from_full_res=wl['sample_from_full_res'], padding_mode=wl['padding_mode'])

    def step():
        if clustering:
            ploss, delta = api.gangealing_cluster_loss(gen, stn, ll, loss_fn, resize, 0.5, wl['batch'], 512, False,
                                                       wl['num_heads'], wl['flips'], device, **common)
        else:
            ploss, delta = api.gangealing_loss(gen, stn, ll, loss_fn, resize, 0.5, wl['batch'], 512, False, device,
                                               **common)
        total = ploss
        if 'flow' in wl['transform']:
            total = ploss + tv_w * api.total_variation_loss(delta) + id_w * api.flow_identity_loss(delta)
        stn.zero_grad()
        ll.zero_grad()
        total.backward()
        t_optim.step()
        ll_optim.step()
        api.accumulate(ema, stn, 0.5 ** (32 / 10000))
        return ploss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ploss = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    loss = float(ploss.detach())
    assert loss == loss and abs(loss) != float('inf'), 'non-finite loss'
    root = api.root
    return dict(elapsed=elapsed, images=wl['batch'] * steps, loss=loss,
                source=os.path.relpath(root, REPO) if root.startswith(REPO) else root)


def recorded_parity(test, mode):
    """The error of arithmetic `mode` against the reference's fixture `test`, as the GPU suite MEASURED it
    (tests/test_gpu_configs.py -> gpurun_out/parity_report.json -> committed as profiles/parity_rNN.json).  Recorded, not
    measured by this run: the fixture comparison belongs to the test suite."""
    for name in ('parity_r06.json', 'parity_r05.json'):
        try:
            with open(os.path.join(REPO, 'profiles', name)) as f:
                rec = json.load(f).get(test, {}).get(mode)
        except (OSError, ValueError):
            continue
        if rec:
            acts = {k: v['max_abs_err'] for k, v in rec.items()
                    if isinstance(v, dict) and 'max_abs_err' in v and k.endswith(('_first', '_sub'))}
            if acts:
                worst = max(acts, key=acts.get)
                return {'max_abs_err': acts[worst], 'max_abs_err_tensor': f'{test}/{worst}',
                        'max_abs_err_source': f'recorded: profiles/{name} (activations of {test} against the reference\'s CPU '
                                              f'fixture, measured by tests/test_gpu_configs.py; NOT a parity mode - the '
                                              f'1e-4 bound is met by fp16x3)'}
    return {'max_abs_err': None, 'max_abs_err_source': 'not recorded'}


def allreduce_only(device, wl, world, iters, warmup):
    """The step's only data-path collective by itself: all-reduce (SUM) of a float32 buffer with the element count of the
    workload's STN gradient arena, `iters` times back to back, each timed with HIP events on the issuing stream (the
    collective runs on RCCL's stream; the events bracket the hand-off there and back).  bus bandwidth = algorithm
    bandwidth x 2 (world - 1) / world (ring all-reduce moves that multiple of the buffer over each link).  On a 7-link
    xGMI node a ring is bound by one link direction (~ 50 - 60 GB/s effective per direction per link): the 172 MB arena
    should take ~ 5 - 6 ms at 8 ranks if RCCL uses one ring, ~ 1 ms with all links - compare with `distributed.
    allreduce_exposed_ms_per_step_max_rank` of the full run."""
    import torch.distributed as dist
    from gangealing_amd.spatial_transformers.spatial_transformer import get_stn
    with torch.device('meta'):
        stn = get_stn(list(wl['transform']), flow_size=wl['flow_size'], supersize=wl['gen_size'], channel_multiplier=0.5,
                      num_heads=wl['num_heads'])
    numel = sum(p.numel() for p in stn.parameters())
    buf = torch.randn(numel, device=device) * 1e-3
    have_group = dist.is_available() and dist.is_initialized()

    def reduce_():
        if have_group:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)

    for _ in range(max(warmup, 2)):
        reduce_()
    torch.cuda.synchronize()
    if have_group:
        dist.barrier()
    evs = []
    t0 = time.perf_counter()
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        reduce_()
        b.record()
        evs.append((a, b))
        buf.mul_(1.0 / max(world, 1))          # keep the values bounded; also forces the stream hand-off each iteration
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    mine = torch.tensor([sum(ms) / len(ms), ms[0], ms[-1], 1e3 * wall / iters], device=device, dtype=torch.float64)
    rows = [mine]
    if have_group and world > 1:
        rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
    per_rank = [[round(float(v), 4) for v in r.tolist()] for r in rows]
    mean_ms = max(r[0] for r in per_rank)
    nbytes = numel * 4
    algbw = nbytes / (mean_ms * 1e-3) / 1e9 if mean_ms > 0 else 0.0
    return {'metric': 'gradient all-reduce of the STN arena (bench.py --allreduce-only)', 'n_gpus': world,
            'backend': dist.get_backend() if have_group else 'none (single process: no collective issued)',
            'bytes': nbytes, 'iters': iters,
            'ms_mean_max_over_ranks': round(mean_ms, 4), 'algbw_GBps': round(algbw, 2),
            'busbw_GBps': round(algbw * 2 * (world - 1) / max(world, 1), 2),
            'per_rank_ms_mean_min_max_wall': per_rank,
            'collective_env': {k: v for k, v in sorted(os.environ.items())
                               if k.startswith(('NCCL_', 'RCCL_', 'HSA_', 'TORCH_NCCL_')) or k == 'HIP_VISIBLE_DEVICES'}}


def roofline_of(run, precision, workload, batch):
    """The `roofline` object of one measured run (measure()): the workload's dominant kernel, timed with HIP events over
    the timed region on the stream it is launched on.  `kernel` is what the library's dispatcher REPORTED for the timed
    launches (gg_last_conv_kernel), never a name kept in this file: a dispatch change shows up here."""
    psum = run['prof']
    if psum is None:
        return None
    if run.get('dominant'):
        # another workload than C2 at its benchmark batch: the kernel the survey found on top
        name, unit = run['dominant'], run['dominant_unit']
        rate = psum['total_flops'] / (psum['total_ms'] * 1e-3) / 1e12 if psum['total_ms'] > 0 else 0.0
        if unit == 'byte':
            bound, peak, u = 'hbm', 8.0, 'TB/s'
        else:
            bound, peak, u = 'mfma', (MFMA_PEAK_TFLOPS['fp32'] if 'fp32' in name else MFMA_PEAK_TFLOPS[precision]), 'TFLOP/s'
        roof = {
            'bound': bound, 'achieved': round(rate, 3), 'peak': peak, 'unit': u, 'frac': round(rate / peak, 4),
            'traffic': pmc_traffic(precision, workload, batch),
            'traffic_source': pmc_traffic_source(precision, workload, batch), 'kernel': name,
            'note': 'the kernel instantiation with the largest share of this workload\'s GPU time (survey below: two '
                    'un-timed steps with HIP events around every convolution / FIR launch, keyed by '
                    'gg_last_conv_kernel); achieved = algorithmic FLOPs (2*N*Cin*Cout*k*k*positions) or algorithmic '
                    'bytes (4 B per input and output element) of its launches / their HIP-event time over the timed '
                    'region; peak = dense MFMA peak of the instruction used / 8 TB/s HBM',
            'launches': psum['launches'],
            'avg_launch_ms': round(psum['total_ms'] / max(psum['launches'], 1), 4),
        }
        if unit != 'byte' and 'fp32' not in name:
            roof['mfma_products_per_flop'] = MFMA_PRODUCTS[precision]
            roof['mfma_issue_frac'] = round(rate * MFMA_PRODUCTS[precision] / peak, 4)
    else:
        achieved = psum['total_flops'] / (psum['total_ms'] * 1e-3) / 1e12 if psum['total_ms'] > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[precision]
        reported = run.get('reported') or []
        roof = {
            'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak,
            'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
            'traffic': pmc_traffic(precision, workload, batch),
            'traffic_source': pmc_traffic_source(precision, workload, batch),
            'kernel': ' + '.join(reported) if reported else 'unknown',
            'kernel_detail': KERNEL_NAME[precision],
            'mfma_products_per_flop': MFMA_PRODUCTS[precision],
            'mfma_issue_frac': round(achieved * MFMA_PRODUCTS[precision] / peak, 4),
            'note': 'kernel = the instantiation(s) the dispatcher reported (gg_last_conv_kernel) for the timed launches: '
                    'the generator\'s style-scaled 3x3 stride-1 layers that fill the chip; achieved = algorithmic conv '
                    'FLOPs (2*N*Cin*Cout*9*OH*OW per launch) / HIP-event time over '
                    'the timed region; peak = dense MFMA peak of the instruction used; mfma_issue_frac = share of '
                    'that peak the matrix pipe actually executes (split precision issues 3 or 6 MFMA products '
                    'per algorithmic product); traffic = HBM bytes per launch from the committed rocprofv3 PMC '
                    'passes (profiles/*pmc_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), null '
                    'when the run is not the profiled configuration',
            'launches': psum['launches'],
            'avg_launch_ms': round(psum['total_ms'] / max(psum['launches'], 1), 4),
            'avg_launch_gflop': round(psum['total_flops'] / max(psum['launches'], 1) / 1e9, 3),
        }
    if run.get('survey'):
        roof['kernels'] = run['survey']
    sc = run.get('step_classes')
    if sc:
        ms_step = 1e3 * run['elapsed'] / max(run.get('steps', 1), 1)
        tflop = sc['mfma_flop'] / 1e12
        roof['step'] = {
            'algorithmic_tflop_per_step': round(tflop, 4),
            'achieved_tflops': round(tflop / (ms_step * 1e-3), 2), 'peak_tflops': MFMA_PEAK_TFLOPS[precision],
            'frac': round(tflop / (ms_step * 1e-3) / MFMA_PEAK_TFLOPS[precision], 4),
            'mfma_class': {'ms_per_step': round(sc['mfma_ms'], 3), 'launches_per_step': sc['mfma_launches'],
                           'tflops': round(tflop / (sc['mfma_ms'] * 1e-3), 2) if sc['mfma_ms'] > 0 else None},
            'hbm_class': {'ms_per_step': round(sc['hbm_ms'], 3), 'launches_per_step': sc['hbm_launches'],
                          'algorithmic_gb_per_step': round(sc['hbm_bytes'] / 1e9, 3),
                          'tb_per_s': round(sc['hbm_bytes'] / 1e12 / (sc['hbm_ms'] * 1e-3), 3) if sc['hbm_ms'] > 0 else None,
                          'frac_of_8_tb_per_s': round(sc['hbm_bytes'] / 1e12 / (sc['hbm_ms'] * 1e-3) / 8.0, 4)
                          if sc['hbm_ms'] > 0 else None},
            'rest_ms_per_step': round(ms_step - sc['mfma_ms'] - sc['hbm_ms'], 3),
            'library_calls_per_step': round(run.get('lib_calls', 0.0), 1),
            'note': 'the WHOLE step: algorithmic convolution FLOPs of every convolution launch of a step (2*N*Cin*Cout*k*k*'
                    'positions, summed over the survey\'s launches; matches SURVEY.md appendix C) / ms_per_step of the timed '
                    'region, as a fraction of the dense 16-bit MFMA peak; mfma_class / hbm_class = HIP-event time of all '
                    'convolution / all FIR-blur launches in the two survey steps (events around every launch: slightly '
                    'slower than the timed region); rest = streaming activations, samplers, perceptual-loss tails, '
                    'optimizer, ATen glue and idle; library_calls_per_step = C-ABI entry-point calls per step in the timed '
                    'region (an entry point launches 1 - 3 kernels; the rocprofv3 dispatch count is in profiles/)'}
    return roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the workload\'s)')
    ap.add_argument('--precision', default=os.environ.get('GANGEALING_CONV_PRECISION', 'fp16x3'),
                    choices=['fp32', 'bf16', 'bf16x3', 'bf16x6', 'fp16x3'],
                    help='arithmetic of the implicit-GEMM convolutions: fp16x3 (default) / bf16x3 = two 16-bit limbs per fp32 '
                         'operand, 3 MFMA products (binary16 limbs on the forward convolutions / bf16 limbs everywhere); '
                         'fp32 = exact-product fp32 MFMA')
    ap.add_argument('--graph', action='store_true',
                    help='time hipGraph replays of the iteration instead of eager launches (one graph on a single GPU; no '
                         'roofline entry: HIP events cannot be recorded inside a replayed graph).  With --gpus > 1 the '
                         'gradient all-reduces are captured inside the graph (nccl / RCCL backend only; gloo raises)')
    ap.add_argument('--allreduce-only', action='store_true',
                    help='micro-mode for attributing a poor scaling curve: no training step, only the gradient exchange of '
                         'one - 50 all-reduces of a buffer the size of the STN gradient arena (C2: 43 M fp32 = 172 MB) on '
                         'the same process group; prints per-rank ms, algorithm and bus bandwidth')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the additional single-GPU measurements (hipGraph replay, plain-bf16 arithmetic)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:           # the child of cpu_baseline_child(): host cores only, no HIP library
        print(json.dumps(cpu_baseline(args.cpu_budget)), flush=True)
        return

    from gangealing_amd import _lib
    _lib.load()                      # fail loudly when the HIP library is missing
    from gangealing_amd import distributed as gdist

    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    # developer switches for exercising the multi-process path on a one-GPU box: all ranks on cuda:0 and a gloo
    # process group (RCCL refuses two ranks on one device); the driver's runs use neither
    if os.environ.get('GANGEALING_SHARE_DEVICE'):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    gdist.setup_distributed(os.environ.get('GANGEALING_DIST_BACKEND', 'nccl'))
    rank = gdist.get_rank()

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl['batch'] = args.batch
    if args.allreduce_only:
        out = allreduce_only(device, wl, world, max(args.steps, 1) * 5, args.warmup)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            gdist.synchronize()
            torch.distributed.destroy_process_group()
        return
    main_run = measure(device, wl, args.precision, args.graph, args.steps, args.warmup, world, gdist)

    if rank == 0:
        elapsed = main_run['elapsed']
        out = {
            'metric': METRIC.get(args.workload, METRIC['c2']),
            'value': round(main_run['images'] / elapsed, 3),
            'unit': 'images/sec',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': DTYPE[args.precision],
            'data': 'synthetic',
            'config': {'workload': f'{args.workload}: gen {wl["gen_size"]}^2, STN {"+".join(wl["transform"])} @ '
                                   f'{wl["flow_size"]}^2, per-GPU batch {wl["batch"]}, VGG16-topology perceptual loss '
                                   f'(random weights), random-init frozen G, psi 0.5, learning rates {SYNTHETIC_LR}',
                       'global_batch': world * wl['batch'], 'parallelism': f'dp{world}',
                       'launch': 'hipGraph replay of the whole iteration' if main_run['graphed'] else 'eager launches',
                       'loss': main_run['loss']},
        }
        if main_run.get('dist'):
            out['distributed'] = main_run['dist']
        roof = roofline_of(main_run, args.precision, args.workload, wl['batch'])
        if roof is not None:
            out['roofline'] = roof
    if world == 1 and rank == 0 and not args.no_extras and not args.graph:
        # further measurements of the same workload in the same process (each with its own trainer); `value` above
        # stays the eager, parity-preserving run whose dominant kernel was timed with HIP events
        extras = {}
        for name, prec, graph, form in (('hipgraph_replay', args.precision, True, 'shared'),
                                        ('bf16x3_eager', 'bf16x3', False, 'shared'),
                                        ('bf16_eager', 'bf16', False, 'shared'),
                                        ('bf16_hipgraph_replay', 'bf16', True, 'shared'),
                                        # the reference's generator FORMULATION (per-sample weights + grouped
                                        # convolutions through op.conv2d_gradfix) inside this package's trainer
                                        ('grouped_form', args.precision, False, 'grouped')):
            if (name.startswith('bf16_') and args.precision == 'bf16') or (name == 'bf16x3_eager' and args.precision == 'bf16x3'):
                continue
            try:
                r = measure(device, wl, prec, graph, args.steps, args.warmup, world, gdist, profile=False, modconv=form)
                extras[name] = {'value': round(r['images'] / r['elapsed'], 3),
                                'ms_per_step': round(1e3 * r['elapsed'] / args.steps, 3), 'dtype': DTYPE[prec],
                                'launch': 'hipGraph replay' if r['graphed'] else 'eager'}
                if prec == 'bf16':
                    extras[name].update(recorded_parity('cfg_c2', 'bf16'))
                if form == 'grouped':
                    extras[name]['generator'] = ('reference formulation: materialised (N*Cout, Cin, k, k) weights, '
                                                 'conv2d / conv_transpose2d with groups = N')
            except Exception as e:             # noqa: BLE001 - an extra must never take the headline line down
                extras[name] = {'error': str(e)[:200]}
        # the reference's own C2 recipe is 8 GPUs x batch 5 (scripts/training/lsun_cats_lpips.sh): the per-GPU step there,
        # eager (bound by the host's launch rate) and as a hipGraph replay (what GangealingTrainer(use_graph='auto') picks
        # at batches <= 8)
        if args.workload == 'c2' and not args.batch:
            for name, graph in (('batch5_eager', False), ('batch5_hipgraph', True)):
                try:
                    r = measure(device, dict(wl, batch=5), args.precision, graph, args.steps, args.warmup, world, gdist,
                                profile=False)
                    extras[name] = {'value': round(r['images'] / r['elapsed'], 3),
                                    'ms_per_step': round(1e3 * r['elapsed'] / args.steps, 3), 'dtype': DTYPE[args.precision],
                                    'launch': 'hipGraph replay' if r['graphed'] else 'eager', 'per_gpu_batch': 5}
                except Exception as e:         # noqa: BLE001
                    extras[name] = {'error': str(e)[:200]}
        # BASELINE configs[3] / [4] at the per-GPU batch of the reference's own 8-GPU recipes (scripts/training/celeba.sh,
        # lsun_cars.sh: 16), each with the roofline of ITS dominant kernel (parity at these batches: tests/golden/
        # cfg_c4b16.npz, cfg_c5b16.npz)
        if args.workload == 'c2' and not args.batch:
            for name, key in (('c4_batch16', 'c4'), ('c5_batch16', 'c5')):
                try:
                    wlx = dict(WORKLOADS[key], batch=16)
                    nsteps = min(args.steps, 10)
                    r = measure(device, wlx, args.precision, False, nsteps, 2, world, gdist, profile=True)
                    extras[name] = {'metric': METRIC[key], 'value': round(r['images'] / r['elapsed'], 3), 'unit': 'images/sec',
                                    'ms_per_step': round(1e3 * r['elapsed'] / nsteps, 3), 'steps': nsteps, 'warmup': 2,
                                    'dtype': DTYPE[args.precision], 'launch': 'eager', 'loss': r['loss'],
                                    'config': {'workload': f'{key}: gen {wlx["gen_size"]}^2, STN @ {wlx["flow_size"]}^2, per-GPU '
                                                           f'batch 16, heads {wlx["num_heads"]}, flips {wlx["flips"]}'},
                                    'roofline': roofline_of(r, args.precision, key, 16)}
                except Exception as e:         # noqa: BLE001
                    extras[name] = {'error': str(e)[:300]}
        # the literal drop-in route: the reference's OWN modules on the HIP operators (last: it binds `models.*`)
        try:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):       # the reference's modules print while they build
                r = measure_reference_dropin(device, wl, args.precision, args.steps, args.warmup)
            if r is None:
                extras['dropin_route'] = {'error': 'no reference Python on this box (oracle/_ref/pyref not staged)'}
            else:
                extras['dropin_route'] = {
                    'value': round(r['images'] / r['elapsed'], 3), 'ms_per_step': round(1e3 * r['elapsed'] / args.steps, 3),
                    'dtype': DTYPE[args.precision], 'launch': 'eager', 'loss': r['loss'],
                    'modules': f'the reference\'s own networks.py / spatial_transformer.py / warping_heads.py / '
                               f'latent_learner.py / loss.py / lpips.py (from {r["source"]}) through '
                               f'gangealing_amd.launch.inject, torch.optim.Adam, models.accumulate'}
        except Exception as e:                 # noqa: BLE001
            extras['dropin_route'] = {'error': str(e)[:300]}
        # ... and the launcher's --modules route: the same training glue (the reference's models/__init__.py,
        # latent_learner.py, torch.optim.Adam, models.accumulate), this package's generator / STN / loss modules bound
        # under the reference's module names (shared-weight modulated convolutions instead of groups = N)
        try:
            with contextlib.redirect_stdout(sys.stderr):
                r = measure_reference_dropin(device, wl, args.precision, args.steps, args.warmup, modules=True)
            if r is not None:
                extras['dropin_modules'] = {
                    'value': round(r['images'] / r['elapsed'], 3), 'ms_per_step': round(1e3 * r['elapsed'] / args.steps, 3),
                    'dtype': DTYPE[args.precision], 'launch': 'eager', 'loss': r['loss'],
                    'modules': 'python -m gangealing_amd.launch --modules: gangealing_amd.stylegan2.networks / '
                               'spatial_transformers / losses under the reference\'s module names; the reference\'s own '
                               f'models/__init__.py, latent_learner.py (from {r["source"]}), torch.optim.Adam, models.accumulate'}
        except Exception as e:                 # noqa: BLE001
            extras['dropin_modules'] = {'error': str(e)[:300]}
        out['extras'] = extras
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline_child(args.cpu_budget)
            except Exception as e:             # noqa: BLE001 - report, never lose the line
                out['cpu_baseline'] = {'error': str(e)[:300]}
        print(json.dumps(out), flush=True)
    if world > 1:
        gdist.synchronize()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
