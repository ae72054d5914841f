"""psi annealing and the learning-rate schedule of the training loop (utils/annealing.py), as plain host
arithmetic: the optimiser here is one fused kernel over a flat arena (csrc/optim.hip) that takes the learning rate
as an argument, so the schedule is a function of the iteration instead of a torch.optim `_LRScheduler`.

  * get_psi_annealing_fn / cosine_anneal / linear_anneal / fastslow_anneal   utils/annealing.py:7-41
  * lr_cycle_iters                                                         utils/annealing.py:44-51
  * DecayingCosineAnnealingWarmRestarts                                    utils/annealing.py:54-146
    (`step(epoch)` with the fractional epoch the training loop passes, train.py:129-132; state_dict keys as the
    reference scheduler's so that `t_sched` / `ll_sched` entries of a reference checkpoint load)
"""
import math


def cosine_anneal(i, maxval, minval, num_steps):
    return minval + 0.5 * (maxval - minval) * (1.0 + math.cos(math.pi * i / num_steps))


def linear_anneal(i, maxval, minval, num_steps):
    return maxval - i * (maxval - minval) / num_steps


def fastslow_anneal(i, maxval, minval, num_steps, a=0.3):
    assert maxval == 1.0 and minval == 0.0
    na = num_steps * a
    return (na - a * i) / (na + i)


def get_psi_annealing_fn(anneal_fn):
    if anneal_fn == 'linear':
        return linear_anneal
    if anneal_fn == 'cosine':
        return cosine_anneal
    raise NotImplementedError(anneal_fn)


def lr_cycle_iters(anneal_psi, period, iter, tm):
    """Iterations at which the learning rate returns to zero (checkpoints are written there, train.py:70-71)."""
    zero_lr_iters = [anneal_psi - 1]
    num_cycles = int(math.log((iter - anneal_psi) / period, tm))
    for n in range(num_cycles):
        zero_lr_iters.append(int(zero_lr_iters[-1] + period * tm ** n))
    return zero_lr_iters


class DecayingCosineAnnealingWarmRestarts:
    """SGDR cosine schedule whose peak decays by `decay` at every restart.  `step(epoch)` follows the reference's
    explicit-epoch branch (utils/annealing.py:112-127), `step()` its implicit one (:105-111)."""

    def __init__(self, base_lr, T_0, decay=0.9, T_mult=1, eta_min=0.0):
        if T_0 <= 0 or not isinstance(T_0, int):
            raise ValueError(f'Expected positive integer T_0, but got {T_0}')
        if T_mult < 1 or not isinstance(T_mult, int):
            raise ValueError(f'Expected integer T_mult >= 1, but got {T_mult}')
        self.base_lrs = [float(base_lr)]
        self.T_0, self.T_i, self.T_mult = T_0, T_0, T_mult
        self.eta_min, self.decay, self.cur_decay = eta_min, decay, 1.0
        self.last_epoch = -1
        self.T_cur = -1
        self._last_lr = [float(base_lr)]
        self.step()                  # _LRScheduler.__init__ performs one initial step (-> epoch 0)

    def get_lr(self):
        return [self.cur_decay * (self.eta_min + (b - self.eta_min) * (1 + math.cos(math.pi * self.T_cur / self.T_i)) / 2)
                for b in self.base_lrs]

    def get_last_lr(self):
        return self._last_lr

    def step(self, epoch=None):
        n = 0
        if epoch is None and self.last_epoch < 0:
            epoch = 0
        if epoch is None:
            epoch = self.last_epoch + 1
            self.T_cur = self.T_cur + 1
            if self.T_cur >= self.T_i:
                self.T_cur = self.T_cur - self.T_i
                self.T_i = self.T_i * self.T_mult
            # (the reference leaves `n` undefined on this branch; it is only reachable through the initial step)
            n = 0 if self.cur_decay == 1.0 else round(math.log(self.cur_decay, self.decay))
        else:
            if epoch < 0:
                raise ValueError(f'Expected non-negative epoch, but got {epoch}')
            if epoch >= self.T_0:
                if self.T_mult == 1:
                    self.T_cur = epoch % self.T_0
                    n = int(epoch // self.T_0)
                else:
                    n = int(math.log((epoch / self.T_0 * (self.T_mult - 1) + 1), self.T_mult))
                    self.T_cur = epoch - self.T_0 * (self.T_mult ** n - 1) / (self.T_mult - 1)
                    self.T_i = self.T_0 * self.T_mult ** n
            else:
                self.T_i = self.T_0
                self.T_cur = epoch
        self.cur_decay = self.decay ** n
        self.last_epoch = math.floor(epoch)
        self._last_lr = self.get_lr()
        return self._last_lr[0]

    _KEYS = ('T_0', 'T_i', 'T_mult', 'eta_min', 'decay', 'cur_decay', 'base_lrs', 'last_epoch', 'T_cur', '_last_lr')

    def state_dict(self):
        return {k: getattr(self, k) for k in self._KEYS}

    def load_state_dict(self, state):
        for k in self._KEYS:
            if k in state:
                setattr(self, k, state[k])
