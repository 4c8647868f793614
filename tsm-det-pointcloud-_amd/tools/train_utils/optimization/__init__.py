"""Optimizer / scheduler builders (reference tools/train_utils/optimization/__init__.py:11-60).

`adam_onecycle` in the reference is fastai's OptimWrapper around Adam(betas=(0.9,0.99)) with true weight decay and a
one-cycle LR / momentum schedule.  Here it is torch.optim.AdamW (decoupled weight decay == fastai `true_wd`; fused
multi-tensor HIP kernel) driven by OneCycle: same LR/momentum curves (cosine up for PCT_START, cosine down after)."""
import math

import torch
import torch.optim as optim


def build_optimizer(model, optim_cfg):
    params = [p for p in model.parameters() if p.requires_grad]
    name = optim_cfg.OPTIMIZER
    if name == 'adam':
        return optim.Adam(params, lr=optim_cfg.LR, weight_decay=optim_cfg.WEIGHT_DECAY)
    if name == 'sgd':
        return optim.SGD(params, lr=optim_cfg.LR, weight_decay=optim_cfg.WEIGHT_DECAY, momentum=optim_cfg.MOMENTUM)
    if name == 'adam_onecycle':
        fused = all(p.is_cuda for p in params)
        return optim.AdamW(params, lr=optim_cfg.LR, betas=(0.9, 0.99), weight_decay=optim_cfg.WEIGHT_DECAY,
                           fused=fused)
    raise NotImplementedError(name)


class OneCycle(object):
    """LR: LR/DIV_FACTOR -> LR over PCT_START of the steps (cosine), then -> LR/(DIV_FACTOR*1e4);
    beta1: MOMS[0] -> MOMS[1] -> MOMS[0] (reference learning_schedules_fastai.py OneCycle)."""

    def __init__(self, optimizer, total_step, lr_max, moms, div_factor, pct_start):
        self.optimizer, self.total_step = optimizer, max(int(total_step), 1)
        self.lr_max, self.moms, self.div_factor, self.pct_start = lr_max, list(moms), div_factor, pct_start
        a1 = int(self.total_step * pct_start)
        self.phases = ((a1, lr_max / div_factor, lr_max, moms[0], moms[1]),
                       (self.total_step - a1, lr_max, lr_max / div_factor / 1e4, moms[1], moms[0]))

    @staticmethod
    def _cos(start, end, pct):
        return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)

    def step(self, it):
        it = min(int(it), self.total_step)
        n1 = self.phases[0][0]
        n, lo, hi, m0, m1 = self.phases[0] if it < n1 else self.phases[1]
        pct = (it if it < n1 else it - n1) / max(n, 1)
        lr, mom = self._cos(lo, hi, pct), self._cos(m0, m1, pct)
        for g in self.optimizer.param_groups:
            g['lr'] = lr
            if 'betas' in g:
                g['betas'] = (mom, g['betas'][1])
        return lr


def build_scheduler(optimizer, total_iters_each_epoch, total_epochs, last_epoch, optim_cfg):
    total_steps = total_iters_each_epoch * total_epochs
    if optim_cfg.OPTIMIZER == 'adam_onecycle':
        return OneCycle(optimizer, total_steps, optim_cfg.LR, list(optim_cfg.MOMS), optim_cfg.DIV_FACTOR,
                        optim_cfg.PCT_START), None
    decay_steps = [x * total_iters_each_epoch for x in optim_cfg.DECAY_STEP_LIST]

    def lr_lbmd(cur_epoch):
        cur_decay = 1
        for d in decay_steps:
            if cur_epoch >= d:
                cur_decay = cur_decay * optim_cfg.LR_DECAY
        return max(cur_decay, optim_cfg.LR_CLIP / optim_cfg.LR)

    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lbmd, last_epoch=last_epoch), None
