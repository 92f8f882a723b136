"""Golden AdamW trajectories for the optimizer row, derived from torch.optim.AdamW.

pytorch-transformers 1.0.0 (the package the reference imports, requirements.txt:1) is not vendored and its AdamW is
gone from the transformers package installed here, so the pin is built from torch.optim.AdamW plus the two ANALYTIC
differences between the algorithms (t = 1-based step, bc1 = 1 - beta1^t, bc2 = 1 - beta2^t, U_t = the Adam update):

  pytorch-transformers:  U_t = lr * sqrt(bc2)/bc1 * m / (sqrt(v) + eps)         [correct_bias=True]
                         U_t = lr * m / (sqrt(v) + eps)                          [correct_bias=False]
                         p_t = (p_{t-1} - U_t) * (1 - lr * wd)                   (decay AFTER the update, on the new value)
  torch.optim.AdamW:     U_t = lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps_torch),     p_t = p_{t-1} * (1 - lr * wd) - U_t

  => with eps_torch = eps / sqrt(bc2) (set before every step) and wd = 0 the two updates are IDENTICAL; U_t does not
     depend on p, so the decayed trajectory follows from the undecayed torch run as p_t = (p_{t-1} - U_t)(1 - lr wd);
     correct_bias=False multiplies U_t by bc1 / sqrt(bc2).

Run in the build container:  python tests/golden/make_adamw_golden.py   -> tests/golden/adamw_trajectory.npz
"""
import math
import os

import numpy as np
import torch

LR, BETAS, EPS, STEPS, N = 1e-2, (0.9, 0.98), 1e-6, 5, 1537


def grads():
    g = torch.Generator().manual_seed(123)
    p0 = torch.randn(N, generator=g, dtype=torch.float64)
    gs = [torch.randn(N, generator=g, dtype=torch.float64) * 0.1 for _ in range(STEPS)]
    return p0, gs


def main():
    p0, gs = grads()
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=LR, betas=BETAS, eps=EPS, weight_decay=0.0)
    updates = []
    for t, g in enumerate(gs, 1):
        opt.param_groups[0]["eps"] = EPS / math.sqrt(1.0 - BETAS[1] ** t)
        before = p.detach().clone()
        p.grad = g.clone()
        opt.step()
        updates.append(before - p.detach())
    out = {}
    for correct_bias in (True, False):
        for wd in (0.0, 0.01):
            q = p0.clone()
            for t, u in enumerate(updates, 1):
                f = 1.0 if correct_bias else (1.0 - BETAS[0] ** t) / math.sqrt(1.0 - BETAS[1] ** t)
                q = (q - u * f) * (1.0 - LR * wd)
            out["p_cb%d_wd%g" % (int(correct_bias), wd)] = q.numpy()
    st = opt.state[p]
    out["exp_avg"], out["exp_avg_sq"] = st["exp_avg"].numpy(), st["exp_avg_sq"].numpy()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "adamw_trajectory.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
