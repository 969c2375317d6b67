"""The training-step oracle against the reference's own autograd (tests/golden/train_text_identity.npz)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import train as ot


def cases():
    z = np.load(os.path.join(GOLDEN, 'train_text_identity.npz'))
    for ci in range(len(z['cases'])):
        for agg in ('sum', 'mean'):
            for loss in ('logits', 'probs'):
                yield z, ci, agg, loss


@pytest.mark.parametrize('ci,agg,loss', [(c, a, l) for _, c, a, l in cases()])
def test_loss_and_text_gradient_match_reference_autograd(ci, agg, loss):
    z = np.load(os.path.join(GOLDEN, 'train_text_identity.npz'))
    tag = f'c{ci}_{agg}_{loss}'
    got_loss, got_grad, got_logits = ot.fs_text_loss_and_grad(
        z[f'c{ci}_feats'], z[f'c{ci}_valid'], z[f'c{ci}_labels'], z[f'c{ci}_text_param'],
        float(z[f'c{ci}_logit_scale']), agg, loss == 'probs')
    np.testing.assert_allclose(got_logits, z[tag + '_logits'], rtol=2e-5, atol=2e-5)
    assert abs(got_loss - float(z[tag + '_loss'])) < 2e-5 * max(1., abs(got_loss))
    g = z[tag + '_grad']
    assert np.abs(got_grad - g).max() < 2e-5 * max(np.abs(g).max(), 1e-3)


def test_adam_matches_torch():
    import torch
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal((7, 5))
    p = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.Adam([p], lr=3e-3, weight_decay=0.01)
    q, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(1, 6):
        g = rng.standard_normal(p0.shape)
        p.grad = torch.tensor(g)
        opt.step()
        ot.adam_step(q, g, m, v, step, 3e-3, weight_decay=0.01)
        np.testing.assert_allclose(q, p.detach().numpy(), rtol=1e-12, atol=1e-12)


def test_cosine_warmup_shape():
    lr = [ot.cosine_warmup_lr(s, 100, 1e-3, 1e-5, 10) for s in range(101)]
    assert lr[0] == 1e-5 and abs(lr[10] - 1e-3) < 1e-12 and abs(lr[100] - 1e-5) < 1e-12
    assert all(a <= b + 1e-15 for a, b in zip(lr[:10], lr[1:11])) and all(a >= b - 1e-15 for a, b in zip(lr[10:100], lr[11:]))


def test_lr_schedule_equals_the_published_scheduler_driven_by_torch():
    """The reference's schedule (method.py:82-98) is nerv's CosineAnnealingWarmupRestarts, un-vendored;
    it is the widely published scheduler of that name (one cycle here: first_cycle_steps = total steps,
    cycle_mult = 1, gamma = 1, lr initialised to min_lr, stepped once per optimiser step).  Its
    published rule, stated here independently as a torch LRScheduler and driven by a real optimiser
    the way nerv drives it (scheduler.step() after every optimizer.step()), must give the learning
    rate every training step of eventclip_amd.train uses -- fractional warm-up lengths included, as
    `warmup_steps_pct * total_steps` produces them."""
    import math
    import torch
    from eventclip_amd.train import cosine_warmup_lr

    class Published(torch.optim.lr_scheduler.LRScheduler):
        def __init__(self, optimizer, first_cycle_steps, max_lr, min_lr, warmup_steps):
            self.first_cycle_steps, self.max_lr, self.min_lr = first_cycle_steps, max_lr, min_lr
            self.warmup_steps, self.step_in_cycle = warmup_steps, -1
            for g in optimizer.param_groups:              # init_lr(): every group starts at min_lr
                g['lr'] = min_lr
            self.base = [min_lr for _ in optimizer.param_groups]
            super().__init__(optimizer)                    # steps once: step_in_cycle = 0

        def get_lr(self):
            if self.step_in_cycle == -1:
                return self.base
            if self.step_in_cycle < self.warmup_steps:
                return [(self.max_lr - b) * self.step_in_cycle / self.warmup_steps + b for b in self.base]
            return [b + (self.max_lr - b) * (1 + math.cos(math.pi * (self.step_in_cycle - self.warmup_steps) /
                                                          (self.first_cycle_steps - self.warmup_steps))) / 2
                    for b in self.base]

        def step(self, epoch=None):
            self.step_in_cycle += 1
            if self.step_in_cycle >= self.first_cycle_steps:
                self.step_in_cycle -= self.first_cycle_steps
            for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
                g['lr'] = lr

    for total, pct, lr in ((100, 0.05, 1e-3), (37, 0.05, 5e-4), (250, 0.1, 2e-2), (12, 0.0, 1e-3)):
        p = torch.nn.Parameter(torch.zeros(3))
        opt = torch.optim.Adam([p], lr=lr)
        sched = Published(opt, total, lr, lr / 100., pct * total)
        for step in range(total):
            used = opt.param_groups[0]['lr']               # what optimizer.step() number `step` runs at
            mine = cosine_warmup_lr(step, total, lr, lr / 100., pct * total)
            assert abs(used - mine) <= 1e-12 * lr, (total, pct, step, used, mine)
            p.grad = torch.ones(3)
            opt.step()
            sched.step()


def trans_cases():
    z = np.load(os.path.join(GOLDEN, 'train_text_trans.npz'))
    return [(ci, agg, loss) for ci in range(len(z['cases'])) for agg, loss in (('sum', 'logits'), ('mean', 'probs'))]


@pytest.mark.parametrize('ci,agg,loss', trans_cases())
def test_text_trans_gradients_match_reference_autograd(ci, agg, loss):
    z = np.load(os.path.join(GOLDEN, 'train_text_trans.npz'))
    tag = f'c{ci}_{agg}_{loss}'
    sd = {k[2:]: z[k] for k in z.files if k.startswith('w:')}
    got_loss, grads, logits = ot.fs_trans_loss_and_grads(
        sd, z[f'c{ci}_feats'], z[f'c{ci}_valid'], z[f'c{ci}_labels'], z[f'c{ci}_text_param'],
        float(z[f'c{ci}_logit_scale']), int(z['adcfg_num_heads']), float(z['adcfg_residual']), agg, loss == 'probs')
    assert abs(got_loss - float(z[tag + '_loss'])) < 1e-4 * max(1., abs(got_loss))
    np.testing.assert_allclose(logits, z[tag + '_logits'], rtol=1e-4, atol=1e-4)
    assert set(grads) == {k.split('_g:')[1] for k in z.files if k.startswith(tag + '_g:')}
    for k, g in grads.items():
        want = z[f'{tag}_g:{k}']
        assert np.abs(g - want).max() < 2e-4 * max(np.abs(want).max(), 1e-3), k
