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
                if p.dim() == 4:         # conv weights: the backward adds straight into the arena (conv_mfma.GRAD_SLOTS)
                    conv_mfma.GRAD_SLOTS[p.data_ptr()] = p.grad
                off += n
        self.params = params
        self.step_count = 0

    def release(self):
        """Forget the gradient slots of this arena (the registry would otherwise keep the arena alive)."""
        for p in getattr(self, 'params', ()):
            slot = conv_mfma.GRAD_SLOTS.get(p.data_ptr())
            if slot is not None and slot.data_ptr() >= self.grad.data_ptr() and \
                    slot.data_ptr() < self.grad.data_ptr() + 4 * self.numel:
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
                if p.dim() == 4:
                    conv_mfma.GRAD_SLOTS[p.data_ptr()] = p.grad
            off += n


def adam_ema_step(arena, lr, ema_flat=None, ema_decay=0.0, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
    arena.step_count += 1
    _lib.call('gg_adam_ema_f32', arena.param, arena.exp_avg, arena.exp_avg_sq, ema_flat, arena.grad, arena.numel,
              lr, betas[0], betas[1], eps, arena.step_count, ema_decay, grad_scale)
    # the kernel wrote through raw pointers: tell autograd's version counters (shared by the parameter views), so
    # that anything cached against the old values - e.g. conv_mfma's weight packs - is recognised as stale
    torch.autograd.graph.increment_version(arena.param)
    if ema_flat is not None:
        torch.autograd.graph.increment_version(ema_flat)


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
                 loss_fn='vgg_ssl', seed=0, perturb_heads=0.0, pipeline_update=None):
        self.device = device
        self._pending = None
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
        self.loss_fn = get_perceptual_loss(loss_fn, device)
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
        self.pipeline_update = (world > 1) if pipeline_update is None else bool(pipeline_update)
        self.stn.register_forward_pre_hook(lambda module, inputs: self.flush())

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

    def step(self, psi=0.5):
        """One iteration: loss forward, backward, gradient all-reduce, Adam x2, EMA (train.py:106-134).

        With `pipeline_update` (default: on when world > 1) the STN half of the update is deferred: the 172 MB
        gradient all-reduce is started asynchronously after backward and the STN's Adam + EMA + weight re-pack run
        right before the STN is next used (a forward pre-hook) - i.e. behind the next iteration's two generator
        passes, which do not read the STN.  Parameter values seen by every forward are exactly those of the
        un-pipelined order; call `flush()` before reading parameters from outside (checkpoints, evaluation)."""
        if self._pending is None:
            self.stn_arena.zero_grad()
        self.ll_arena.zero_grad()
        total, parts = self.loss(psi)            # (its STN forward first applies a pending update and zeroes the grads)
        total.backward()
        scale = 1.0 / self.world
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.ll_arena.grad, op=dist.ReduceOp.SUM)
            work = dist.all_reduce(self.stn_arena.grad, op=dist.ReduceOp.SUM, async_op=self.pipeline_update)
        else:
            work = None
        if not self.freeze_ll:
            adam_ema_step(self.ll_arena, self.ll_lr, grad_scale=scale)
        if self.pipeline_update:
            self._pending = (work, scale)
        else:
            self._apply_stn_update(scale)
        return parts

    def _apply_stn_update(self, scale):
        adam_ema_step(self.stn_arena, self.stn_lr, self.ema_arena.param, self.ema_decay, grad_scale=scale)
        conv_mfma.repack_trainable()         # all STN weight packs (forward + data-gradient layouts) in one launch

    def flush(self):
        """Apply a deferred STN update (no-op when nothing is pending)."""
        if self._pending is not None:
            work, scale = self._pending
            self._pending = None
            if work is not None:
                work.wait()                      # the compute stream waits for the collective; the host does not
            self._apply_stn_update(scale)
            self.stn_arena.zero_grad()


def smoke(device):
    """Tiny end-to-end iteration (used by __graft_entry__.smoke)."""
    tr = GangealingTrainer(device, gen_size=64, flow_size=64, batch=2, transform=('similarity', 'flow'), inject=3,
                           ndirs=2, perturb_heads=0.02)
    before = tr.stn_arena.param.clone()
    parts = tr.step(psi=0.5)
    torch.cuda.synchronize()
    assert torch.isfinite(parts['p']).all() and torch.isfinite(tr.stn_arena.grad).all()
    assert float((tr.stn_arena.param - before).abs().max()) > 0
    return parts
