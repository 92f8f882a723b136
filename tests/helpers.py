"""Shared comparison helpers for the parity tests."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import cases  # noqa: E402

# Parity bar from BASELINE.json north_star: fp32 1e-4. Checked as |got - want| <= ATOL + RTOL * |want|.
# vision_logit carries +(-10000) on masked regions where one fp32 ulp is 9.8e-4 and the reference
# itself is only good to ~5e-4 against fp64; the relative term covers those entries.
ATOL = 1e-4
RTOL = 1e-4


def assert_close(got, want, name="", atol=ATOL, rtol=RTOL):
    got = torch.as_tensor(np.asarray(got.detach().cpu()) if isinstance(got, torch.Tensor) else got).double()
    want = torch.as_tensor(np.asarray(want.detach().cpu()) if isinstance(want, torch.Tensor) else want).double()
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, tuple(got.shape), tuple(want.shape))
    assert torch.isfinite(got).all(), "%s: non-finite values" % name
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    worst = (err - bound).max().item()
    assert worst <= 0, "%s: max abs err %.3e exceeds %.1e + %.1e*|ref| (max |ref| %.3e)" % (
        name, err.max().item(), atol, rtol, want.abs().max().item())
    return err.max().item()


def load_golden(case):
    with np.load(cases.path(case)) as z:
        return {k: z[k] for k in z.files}


def to_device(args, device):
    return tuple(a.to(device) if isinstance(a, torch.Tensor) else a for a in args)
