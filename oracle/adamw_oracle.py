"""CPU restatement of ``pytorch_transformers.optimization.AdamW.step`` (TEST INFRASTRUCTURE ONLY).

The reference pins ``pytorch-transformers==1.0.0`` (/root/reference/requirements.txt:1) and calls
``AdamW(params, lr, betas=(0.9, 0.98))`` (train_concap.py:465-470) / ``AdamW(params, lr, correct_bias=False)``
(train_tasks.py:426). The package is not vendored in the reference tree and not installed here, so this is a
restatement of its published algorithm (one torch op per line of the original ``step``). The real package cannot be
run here; the restatement is PINNED INDIRECTLY: tests/golden/adamw_trajectory.npz holds 5-step trajectories derived
from torch.optim.AdamW through the two analytic differences of the algorithms (eps placement under bias correction,
decay before / after the update - tests/golden/make_adamw_golden.py), for both correct_bias settings, with and
without decay; tests/test_optim.py checks this file and, independently, the native kernel against them, and the
warm-up schedules against the transformers package installed in the image.
"""
import math

import torch


def adamw_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
    """In-place update of fp32/fp64 CPU tensors p, m, v with gradient g at 1-based step count `step`."""
    b1, b2 = betas
    m.mul_(b1).add_(g, alpha=1.0 - b1)                 # exp_avg.mul_(beta1).add_(1.0 - beta1, grad)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)          # exp_avg_sq.mul_(beta2).addcmul_(1.0 - beta2, grad, grad)
    denom = v.sqrt().add_(eps)                         # denom = exp_avg_sq.sqrt().add_(group['eps'])
    step_size = lr
    if correct_bias:                                   # no bias correction for the original BERT recipe
        step_size = step_size * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
    p.addcdiv_(m, denom, value=-step_size)             # p.data.addcdiv_(-step_size, exp_avg, denom)
    if weight_decay > 0.0:                             # decoupled decay AFTER the Adam update
        p.add_(p, alpha=-lr * weight_decay)            # p.data.add_(-group['lr'] * group['weight_decay'], p.data)
    return p


def warmup_linear(step, warmup_steps, t_total):
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))
