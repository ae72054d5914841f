"""One GANgealing training iteration (train.py:89-134) on the HIP operators.

MI355X-specific structure (everything else follows the reference loop):
  * STN parameters, gradients, Adam moments and the EMA copy live in FLAT fp32 arenas; module
    parameters are views into them.  The optimiser step + EMA (train.py:126-134: per-tensor Adam and
    a 130-launch Python EMA loop) is one fused streaming kernel (csrc/optim.hip).
  * data parallel = one process per GPU; gradients are averaged with ONE all-reduce over the flat
    gradient arena (RCCL over xGMI) - no per-bucket hooks, no collective anywhere else in the step.
  * no host synchronisation inside the step: losses stay on the device (the reference calls
    .item() three times per step on rank 0, train.py:140-142).
"""
import math

import torch
import torch.nn as nn

from . import _lib
from .op import conv_mfma
from . import distributed as gdist
from .annealing import DecayingCosineAnnealingWarmRestarts, get_psi_annealing_fn
from .latent_learner import DirectionInterpolator
from .losses import gangealing_loss, gangealing_cluster_loss, flow_losses, get_perceptual_loss
from .spatial_transformers.antialiased_sampling import BilinearDownsample
from .spatial_transformers.spatial_transformer import get_stn
from .stylegan2.networks import Generator


class FlatArena:
    """Re-homes the parameters of `module` into one contiguous fp32 buffer (plus matching gradient and
    Adam-moment buffers).  Parameter tensors become views, so autograd accumulates straight into the
    gradient arena and one kernel / one collective can touch everything."""

    def __init__(self, module):
        params = [p for p in module.parameters()]
        # parameters whose backward can add straight into the arena (conv_mfma.grad_slots): trainable convolution
        # weights and the biases of FusedLeakyReLU activations - nothing else is registered (other 1-D parameters get
        # their gradient from autograd as usual)
        from .op.fused_act import FusedLeakyReLU
        from .stylegan2.networks import EqualConv2d
        # (EqualConv2d biases: the flow head's conv + ReLU pairs run as conv3x3_bias_act, whose backward looks the bias up
        # here; a bias whose layer takes another route simply never asks)
        self._slot_ids = {id(m.bias) for m in module.modules()
                          if isinstance(m, (FusedLeakyReLU, EqualConv2d)) and getattr(m, 'bias', None) is not None}
        self._slot_ids |= {id(p) for p in params if p.dim() == 4}
        self.numel = sum(p.numel() for p in params)
        dev = params[0].device
        self.param = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.param[off:off + n].copy_(p.reshape(-1))
                p.data = self.param[off:off + n].view(p.shape)
                p.grad = self.grad[off:off + n].view(p.shape)
                if id(p) in self._slot_ids and p.requires_grad:
                    # inside `with conv_mfma.grad_slots():` the backward adds straight into the arena
                    conv_mfma.register_grad_slot(p, p.grad)
                off += n
        self.params = params
        self.step_count = 0

    def release(self):
        """Forget the gradient slots of this arena (the registry would otherwise keep the arena alive)."""
        for p in getattr(self, 'params', ()):
            ent = conv_mfma.GRAD_SLOTS.get(p.data_ptr())
            if ent is not None and ent[0].data_ptr() >= self.grad.data_ptr() and \
                    ent[0].data_ptr() < self.grad.data_ptr() + 4 * self.numel:
                del conv_mfma.GRAD_SLOTS[p.data_ptr()]

    def __del__(self):
        try:
            self.release()
        except Exception:       # interpreter shutdown
            pass

    def zero_grad(self):
        self.grad.zero_()
        off = 0
        for p in self.params:          # autograd may have replaced .grad; re-point it at the arena
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + n].view(p.shape)
                if id(p) in self._slot_ids and p.requires_grad:
                    conv_mfma.register_grad_slot(p, p.grad)
            off += n

    def touch(self):
        """Tell autograd that the arena was rewritten through raw pointers.  `p.data = view` gave every parameter its
        OWN version counter (bumping the arena tensor's does not reach them), and everything cached against parameter
        values - conv_mfma's weight packs, EqualLinear's scaled weights - is keyed on those counters."""
        torch.autograd.graph.increment_version(self.params)
        torch.autograd.graph.increment_version(self.param)

    # ---- torch.optim.Adam-compatible optimiser state (train.py:22-28 stores t_optim / ll_optim state_dicts) ----
    def optim_state_dict(self, lr, betas=(0.9, 0.999), eps=1e-8):
        state, off = {}, 0
        for i, p in enumerate(self.params):
            n = p.numel()
            state[i] = {'step': torch.tensor(float(self.step_count)),
                        'exp_avg': self.exp_avg[off:off + n].view(p.shape).clone(),
                        'exp_avg_sq': self.exp_avg_sq[off:off + n].view(p.shape).clone()}
            off += n
        group = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                     capturable=False, differentiable=False, fused=None, params=list(range(len(self.params))))
        return {'state': state, 'param_groups': [group]}

    def load_optim_state_dict(self, sd):
        """Accepts the state_dict of a torch.optim.Adam over the same parameter list (reference checkpoints).
        -> the stored learning rate."""
        groups = sd['param_groups']
        order = [i for g in groups for i in g['params']]
        if len(order) != len(self.params):
            raise ValueError(f'optimizer state has {len(order)} parameters, the arena {len(self.params)}')
        steps, off = set(), 0
        with torch.no_grad():
            for idx, p in zip(order, self.params):
                n = p.numel()
                st = sd['state'].get(idx)
                if st is None:                       # parameter never stepped
                    self.exp_avg[off:off + n].zero_()
                    self.exp_avg_sq[off:off + n].zero_()
                else:
                    if tuple(st['exp_avg'].shape) != tuple(p.shape):
                        raise ValueError(f'optimizer state {idx}: shape {tuple(st["exp_avg"].shape)} vs {tuple(p.shape)}')
                    self.exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1))
                    self.exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                    steps.add(int(float(st['step'])))
                off += n
        if len(steps) > 1:
            raise ValueError(f'per-parameter step counts differ ({sorted(steps)}): the fused kernel keeps one')
        self.step_count = steps.pop() if steps else 0
        return groups[0]['lr']


def adam_hyper(lr, step, betas=(0.9, 0.999), grad_scale=1.0):
    """The step-dependent scalars of the update: [lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale]."""
    return [float(lr), 1.0 - betas[0] ** step, math.sqrt(1.0 - betas[1] ** step), float(grad_scale)]


def adam_ema_step(arena, lr, ema_arena=None, ema_decay=0.0, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, hyper=None):
    """Adam over `arena` and (optionally) the EMA update of `ema_arena` (a FlatArena of the same layout) in one
    streaming kernel.  hyper: device tensor holding adam_hyper(...) - the launch then carries no step-dependent
    argument and can be replayed from a captured graph (`lr` / `grad_scale` are ignored)."""
    arena.step_count += 1
    if hyper is not None:
        _lib.call('gg_adam_ema_dev_f32', arena.param, arena.exp_avg, arena.exp_avg_sq,
                  None if ema_arena is None else ema_arena.param, arena.grad, arena.numel, hyper, betas[0], betas[1],
                  eps, ema_decay)
    else:
        _lib.call('gg_adam_ema_f32', arena.param, arena.exp_avg, arena.exp_avg_sq,
                  None if ema_arena is None else ema_arena.param, arena.grad, arena.numel,
                  lr, betas[0], betas[1], eps, arena.step_count, ema_decay, grad_scale)
    # the kernel wrote through raw pointers: bump the version counters of the parameters themselves, so that
    # anything cached against the old values (weight packs, scaled EqualLinear weights) is recognised as stale
    arena.touch()
    if ema_arena is not None:
        ema_arena.touch()


def cosine_psi(step, total):
    """psi annealing 1 -> 0 (utils/annealing.py cosine schedule)."""
    if step >= total:
        return 0.0
    return 0.5 * (1.0 + math.cos(math.pi * step / total))


class GangealingTrainer:
    """Holds G (frozen), the STN, its EMA copy, the latent learner and the perceptual loss, and runs
    train iterations.  Arguments mirror utils/base_argparse.py:4-60."""

    def __init__(self, device, gen_size=256, flow_size=128, real_size=None, batch=16, transform=('similarity', 'flow'),
                 num_heads=1, flips=False, dim_latent=512, n_mlp=8, gen_channel_multiplier=2,
                 stn_channel_multiplier=0.5, inject=5, ndirs=1, padding_mode='reflection', tv_weight=1000.0,
                 flow_identity_weight=1.0, stn_lr=1e-3, ll_lr=1e-2, sample_from_full_res=False, freeze_ll=False,
                 loss_fn='vgg_ssl', seed=0, perturb_heads=0.0, pipeline_update=None, anneal_psi=150000,
                 anneal_fn='cosine', period=37500, decay=0.9, tm=2, perceptual_weights=None, use_graph=False,
                 graph_warmup=3, allow_random_loss=None, perceptual_trunk_weights=None, collectives=None):
        """allow_random_loss: run on a seeded RANDOM perceptual trunk when the weight files are missing (synthetic
        benchmark / parity runs).  Default None = only when GANGEALING_SYNTHETIC=1 is set; training otherwise raises
        FileNotFoundError instead of silently optimising a meaningless objective.
        collectives: issue the gradient all-reduces (default: when the process group has more than one rank).  True
        on a ONE-rank group runs the exact multi-GPU call sequence - async all-reduce of the flat arena on RCCL's
        stream, work.wait(), deferred Adam / EMA / re-pack - on a single-GPU box (tests/test_gpu_rccl_single_rank.py)."""
        self.device = device
        self._pending = None
        # optional timing of the gradient exchange (bench.py --gpus N): list of (event before, event after) pairs
        # recorded on the compute stream around the point where it waits for the collective - the time the all-reduce
        # is EXPOSED to the step (zero when it finished behind the next iteration's generator passes)
        self.comm_events = None
        self.batch = batch
        self.dim_latent = dim_latent
        self.num_heads = num_heads
        self.flips = flips
        self.padding_mode = padding_mode
        self.tv_weight = tv_weight
        self.flow_identity_weight = flow_identity_weight
        self.stn_lr, self.ll_lr = stn_lr, ll_lr
        self.sample_from_full_res = sample_from_full_res
        self.freeze_ll = freeze_ll
        self.clustering = num_heads > 1
        real_size = gen_size if real_size is None else real_size
        transform = list(transform)
        rank, world = gdist.get_rank(), gdist.get_world_size()
        # identical initial weights on every rank (the reference gets this from DDP's broadcast at wrap time)
        torch.manual_seed(seed)
        self.generator = Generator(gen_size, dim_latent, n_mlp, channel_multiplier=gen_channel_multiplier).to(device)
        self.generator.eval().requires_grad_(False)
        kw = dict(flow_size=flow_size, supersize=real_size, channel_multiplier=stn_channel_multiplier,
                  num_heads=num_heads)
        self.stn = get_stn(transform, **kw).to(device)
        if perturb_heads > 0:      # zero-initialised heads give the identity warp (warping_heads.py:28-30,163-165);
            with torch.no_grad():  # benchmarks perturb them so that mip levels > 0 and real flows are exercised
                for name, p in self.stn.named_parameters():
                    if 'warp_head.linear' in name or 'flow_out.2' in name:
                        p.normal_(0.0, perturb_heads)
        self.t_ema = get_stn(transform, **kw).to(device)
        self.t_ema.load_state_dict(self.stn.state_dict())
        self.t_ema.requires_grad_(False)
        self.ll = DirectionInterpolator(None, ndirs, inject, self.generator.n_latent, num_heads,
                                        dim_latent=dim_latent).to(device)
        if allow_random_loss is None:
            import os
            allow_random_loss = os.environ.get('GANGEALING_SYNTHETIC', '0') == '1'
        self.loss_fn = get_perceptual_loss(loss_fn, device, weights=perceptual_weights,
                                           allow_random=allow_random_loss, trunk_weights=perceptual_trunk_weights)
        self.resize_fake2stn = BilinearDownsample(gen_size // flow_size, 3).to(device) if gen_size > flow_size \
            else nn.Sequential()
        conv_mfma.enable_pack_registry()     # trainable conv weights: packs rebuilt once per step (repack_trainable)
        self.stn_arena = FlatArena(self.stn)
        self.ema_arena = FlatArena(self.t_ema)
        self.ll_arena = FlatArena(self.ll)
        self.ema_decay = 0.5 ** (32 / (10 * 1000))           # train.py:77
        self.is_flow = 'flow' in transform
        # per-rank data stream (train.py:193-194)
        torch.manual_seed(seed * world + rank)
        self.world = world
        # defer the STN optimizer step behind the next iteration's generator passes (see step()); only pays when there
        # is a collective to hide
        self.collectives = (world > 1) if collectives is None else bool(collectives)
        self.pipeline_update = self.collectives if pipeline_update is None else bool(pipeline_update)
        self.stn.register_forward_pre_hook(lambda module, inputs: self._before_stn_forward())
        # hipGraph replay of the iteration: after `graph_warmup` eager iterations the step - zeroing the gradient
        # arenas, loss forward, backward, the gradient all-reduces (when there are ranks to reduce over), both
        # optimizers, EMA, weight re-pack: ~700 launches - is captured ONCE and replayed; psi and the optimizers'
        # step-dependent scalars live in device memory and are refreshed by the host before every replay.
        # With collectives the all-reduces are captured INSIDE the graph: RCCL's collectives are stream-ordered kernel
        # launches on the communicator's stream, forked from / joined to the capturing stream by events, and a hipGraph
        # records exactly that (backend "nccl" only - gloo reduces on the host and cannot be captured).
        # use_graph='auto': replay at per-GPU batches <= 8 in a single process (where the eager step is bound by the
        # host's launch rate), eager launches otherwise (at larger batches the GPU is the bound; with collectives the
        # eager step hides the all-reduce behind the next iteration's generator passes, which one graph cannot).
        # (Round 5 replayed the multi-process step as FOUR graphs with eager collectives between them; that route gave
        # NaN at the second replay in some sessions, the cause was never located, and round 6 removed it.)
        if use_graph == 'auto':
            self.use_graph = batch <= 8 and not self.collectives
        else:
            self.use_graph = bool(use_graph)
        if self.use_graph and self.collectives:
            import torch.distributed as dist
            backend = dist.get_backend() if dist.is_initialized() else None
            if backend != 'nccl':
                raise RuntimeError(f'GangealingTrainer(use_graph=True) with collectives needs the nccl (RCCL) backend: its '
                                   f'all-reduce is captured inside the hipGraph; backend {backend!r} reduces on the host. '
                                   f'Use eager steps (use_graph=False or "auto").')
            self.pipeline_update = False       # one graph per iteration: the update is not deferred
        self._graph = None
        self._graph_calls = 0
        self._graph_warmup = max(int(graph_warmup), 1)
        self._psi_dev = torch.zeros((), dtype=torch.float32, device=device)
        self._hyper_dev = torch.zeros(8, dtype=torch.float32, device=device)        # [stn x4, ll x4]
        self._cap_stream = None
        self._hyper_ring = None
        # psi annealing + learning-rate schedules (train.py:206-207, 92-97, 129-132)
        self.anneal_psi, self.period = anneal_psi, period
        self.anneal_fn = get_psi_annealing_fn(anneal_fn)
        self.t_sched = DecayingCosineAnnealingWarmRestarts(stn_lr, T_0=1, T_mult=tm, decay=decay)
        self.ll_sched = DecayingCosineAnnealingWarmRestarts(ll_lr, T_0=1, T_mult=tm, decay=decay)

    def loss(self, psi):
        common = dict(sample_from_full_res=self.sample_from_full_res, padding_mode=self.padding_mode)
        if self.clustering or self.flips:
            ploss, delta = gangealing_cluster_loss(self.generator, self.stn, self.ll, self.loss_fn,
                                                   self.resize_fake2stn, psi, self.batch, self.dim_latent,
                                                   self.freeze_ll, self.num_heads, self.flips, self.device, **common)
        else:
            ploss, delta = gangealing_loss(self.generator, self.stn, self.ll, self.loss_fn, self.resize_fake2stn, psi,
                                           self.batch, self.dim_latent, self.freeze_ll, self.device, **common)
        total = ploss
        tv = idl = None
        if self.is_flow and (self.tv_weight > 0 or self.flow_identity_weight > 0):
            reg = flow_losses(delta)
            tv, idl = reg[0], reg[1]
            total = total + self.tv_weight * tv + self.flow_identity_weight * idl
        return total, {'p': ploss.detach(), 'tv': tv.detach() if tv is not None else None,
                       'f': idl.detach() if idl is not None else None}

    def train_iteration(self, i):
        """Iteration `i` (1-based, as train.py:89-134 counts): psi from the annealing schedule while i <= anneal_psi,
        then psi = 0 and the two learning-rate schedulers advance by the fractional epoch (i - anneal_psi) / period
        after the optimizer steps.  -> (loss parts, psi)."""
        if i <= self.anneal_psi:
            psi, psi_is_fixed = float(self.anneal_fn(i, 1.0, 0.0, self.anneal_psi)), False
        else:
            psi, psi_is_fixed = 0.0, True
        parts = self.step(psi, stn_lr=self.t_sched.get_last_lr()[0], ll_lr=self.ll_sched.get_last_lr()[0])
        if psi_is_fixed:
            epoch = max(0, (i - self.anneal_psi) / self.period)
            self.t_sched.step(epoch)
            self.ll_sched.step(epoch)
        return parts, psi

    def state_dict(self):
        """Checkpoint in the reference's layout (train.py:22-28; `args` is the caller's to add)."""
        self.flush()
        return {'g_ema': self.generator.state_dict(), 't': self.stn.state_dict(), 't_ema': self.t_ema.state_dict(),
                't_optim': self.stn_arena.optim_state_dict(self.t_sched.get_last_lr()[0]),
                't_sched': self.t_sched.state_dict(), 'll': self.ll.state_dict(),
                'll_optim': self.ll_arena.optim_state_dict(self.ll_sched.get_last_lr()[0]),
                'll_sched': self.ll_sched.state_dict()}

    def load_state_dict(self, ckpt, load_G_only=False):
        """train.py:216-228: `g_ema` always; the rest unless load_G_only / absent (KeyError -> G only)."""
        self.flush()
        self.generator.load_state_dict(ckpt['g_ema'])
        if load_G_only or 't' not in ckpt:
            return False
        self.stn.load_state_dict(ckpt['t'])
        self.t_ema.load_state_dict(ckpt['t_ema'])
        self.stn_arena.load_optim_state_dict(ckpt['t_optim'])
        self.t_sched.load_state_dict(ckpt['t_sched'])
        self.ll.load_state_dict(ckpt['ll'])
        self.ll_arena.load_optim_state_dict(ckpt['ll_optim'])
        self.ll_sched.load_state_dict(ckpt['ll_sched'])
        for arena in (self.stn_arena, self.ema_arena, self.ll_arena):
            arena.touch()                        # copy_ into the views already bumped them; cheap and explicit
        conv_mfma.repack_trainable()
        return True

    def step(self, psi=0.5, stn_lr=None, ll_lr=None):
        """One iteration: loss forward, backward, gradient all-reduce, Adam x2, EMA (train.py:106-134).

        With `pipeline_update` (default: on when world > 1) the STN half of the update is deferred: the 172 MB
        gradient all-reduce is started asynchronously after backward and the STN's Adam + EMA + weight re-pack run
        right before the STN is next used (a forward pre-hook) - i.e. behind the next iteration's two generator
        passes, which do not read the STN.  Parameter values seen by every forward are exactly those of the
        un-pipelined order; call `flush()` before reading parameters from outside (checkpoints, evaluation)."""
        if self.use_graph:
            return self._graph_step(psi, self.stn_lr if stn_lr is None else stn_lr,
                                    self.ll_lr if ll_lr is None else ll_lr)
        if self._pending is None:
            self.stn_arena.zero_grad()
        self.ll_arena.zero_grad()
        total, parts = self.loss(psi)            # (its STN forward first applies a pending update and zeroes the grads)
        with conv_mfma.grad_slots():             # conv weight gradients accumulate straight into the arena
            total.backward()
        scale = 1.0 / self.world
        if self.collectives:
            import torch.distributed as dist
            dist.all_reduce(self.ll_arena.grad, op=dist.ReduceOp.SUM)
            ev = self._comm_mark() if not self.pipeline_update else None
            work = dist.all_reduce(self.stn_arena.grad, op=dist.ReduceOp.SUM, async_op=self.pipeline_update)
            if ev is not None:
                self._comm_mark(ev)
        else:
            work = None
        stn_lr = self.stn_lr if stn_lr is None else stn_lr
        if not self.freeze_ll:
            adam_ema_step(self.ll_arena, self.ll_lr if ll_lr is None else ll_lr, grad_scale=scale)
        if self.pipeline_update:
            self._pending = (work, scale, stn_lr)
        else:
            self._apply_stn_update(scale, stn_lr)
        return parts

    # ---- hipGraph path ------------------------------------------------------------------------------------------
    def _graph_body(self):
        """Everything of one iteration that runs on the GPU, with no step-dependent host scalar in any launch."""
        self.stn_arena.grad.zero_()
        self.ll_arena.grad.zero_()
        total, parts = self.loss(self._psi_dev)
        with conv_mfma.grad_slots():
            total.backward()
        if self.collectives:                     # (captured with the rest: see __init__)
            import torch.distributed as dist
            dist.all_reduce(self.ll_arena.grad, op=dist.ReduceOp.SUM)
            dist.all_reduce(self.stn_arena.grad, op=dist.ReduceOp.SUM)
        if not self.freeze_ll:
            adam_ema_step(self.ll_arena, 0.0, hyper=self._hyper_dev[4:8])
        adam_ema_step(self.stn_arena, 0.0, self.ema_arena, self.ema_decay, hyper=self._hyper_dev[0:4])
        conv_mfma.repack_trainable()
        return parts

    def _graph_step(self, psi, stn_lr, ll_lr):
        # Everything of the graph path - the eager warm-up iterations, the capture, the per-iteration scalar uploads and
        # the replays - runs on ONE dedicated stream; the caller's stream waits for it at the end of every call.
        if self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream()
        self._cap_stream.wait_stream(cur)
        with torch.cuda.stream(self._cap_stream):
            parts = self._graph_step_on_stream(psi, stn_lr, ll_lr)
        cur.wait_stream(self._cap_stream)
        return parts

    def _graph_step_on_stream(self, psi, stn_lr, ll_lr):
        # the scalars this iteration's launches read from device memory (the 1 / world of the gradient average rides in
        # the Adam kernel's grad_scale)
        scale = 1.0 / self.world
        vals = adam_hyper(stn_lr, self.stn_arena.step_count + 1, grad_scale=scale) + \
            adam_hyper(ll_lr, self.ll_arena.step_count + 1, grad_scale=scale)
        # pinned staging ring: the copy is asynchronous (no host sync per iteration); a slot is rewritten 16 iterations
        # later, after its event shows the copy has been consumed
        if self._hyper_ring is None:
            self._hyper_ring = [[torch.zeros(9, dtype=torch.float32).pin_memory(), None] for _ in range(16)]
        slot = self._hyper_ring[self._graph_calls % 16]
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0][:8] = torch.tensor(vals, dtype=torch.float32)
        slot[0][8] = float(psi)
        self._hyper_dev.copy_(slot[0][:8], non_blocking=True)
        self._psi_dev.copy_(slot[0][8], non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        self._graph_calls += 1
        if self._graph is None and self._graph_calls <= self._graph_warmup:
            # eager iterations: every cache, workspace and library handle the capture must not create exists afterwards
            self.stn_arena.zero_grad()           # (re-points .grad at the arenas)
            self.ll_arena.zero_grad()
            return self._graph_body()
        if self._graph is None:
            self._cap_stream.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            steps = (self.stn_arena.step_count, self.ll_arena.step_count)
            with torch.cuda.graph(self._graph, stream=self._cap_stream):
                self._graph_parts = self._graph_body()
            self.stn_arena.step_count, self.ll_arena.step_count = steps       # the capture executed nothing
        self._graph.replay()
        # host-side bookkeeping of what the replay did on the device
        self.stn_arena.step_count += 1
        self.stn_arena.touch()
        self.ema_arena.touch()
        if not self.freeze_ll:
            self.ll_arena.step_count += 1
            self.ll_arena.touch()
        conv_mfma.mark_trainable_packs_current()
        return self._graph_parts

    def _before_stn_forward(self):
        self.flush()

    def _apply_stn_update(self, scale, lr):
        adam_ema_step(self.stn_arena, lr, self.ema_arena, self.ema_decay, grad_scale=scale)
        conv_mfma.repack_trainable()         # all STN weight packs (forward + data-gradient layouts) in one launch

    def _comm_mark(self, start=None):
        if self.comm_events is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if start is not None:
            self.comm_events.append((start, ev))
        return ev

    def flush(self):
        """Apply a deferred STN update (no-op when nothing is pending)."""
        if self._pending is not None:
            work, scale, lr = self._pending
            self._pending = None
            if work is not None:
                ev = self._comm_mark()
                work.wait()                      # the compute stream waits for the collective; the host does not
                self._comm_mark(ev)
            self._apply_stn_update(scale, lr)
            self.stn_arena.zero_grad()


def smoke(device):
    """Tiny end-to-end iteration (used by __graft_entry__.smoke)."""
    tr = GangealingTrainer(device, gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3,
                           ndirs=2, perturb_heads=0.02, allow_random_loss=True)
    before = tr.stn_arena.param.clone()
    parts = tr.step(psi=0.5)
    torch.cuda.synchronize()
    assert torch.isfinite(parts['p']).all() and torch.isfinite(tr.stn_arena.grad).all()
    assert float((tr.stn_arena.param - before).abs().max()) > 0
    return parts
