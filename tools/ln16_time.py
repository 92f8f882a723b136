"""Isolated timings of the bf16 LayerNorm kernels (training: ops16.layernorm_fwd / layernorm_bwd; MX inference: ops.layernorm_fwd
on a bf16 row = vb_layernorm_fwd_mx16) at the step's shapes - HIP events around 50 launches, algorithmic bytes / time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
from vilbert import _native, ops, ops16  # noqa: E402

dev = "cuda:0"


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for rows, cols in ((9216, 768), (9472, 1024), (18432, 768), (18944, 1024), (2304, 768)):
    x = torch.randn(rows, cols, device=dev).to(torch.bfloat16)
    dy = torch.randn(rows, cols, device=dev).to(torch.bfloat16)
    g, b = torch.ones(cols, device=dev), torch.zeros(cols, device=dev)
    y, mean, rstd = ops16.layernorm_fwd(x, g, b, 1e-12, want_stats=True)
    t_f = timed(lambda: ops16.layernorm_fwd(x, g, b, 1e-12, want_stats=True))
    t_b = timed(lambda: ops16.layernorm_bwd(dy, x, mean, rstd, g))
    t_bd = timed(lambda: ops16.layernorm_bwd(dy, x, mean, rstd, g, drop=(0.1, 7)))
    mb = rows * cols * 2 / 1e6
    print("%6d x %4d  fwd %6.1f us (%4.2f TB/s)   bwd %6.1f us (%4.2f TB/s)   bwd + dropped twin %6.1f us (%4.2f TB/s)" % (
        rows, cols, t_f, 2 * mb / t_f, t_b, 3 * mb / t_b, t_bd, 4 * mb / t_bd))
prev = _native.set_gemm_mode("mxfp8")
with torch.no_grad():
    for rows, cols in ((18432, 768), (18944, 1024), (9216, 768)):
        x = torch.randn(rows, cols, device=dev).to(torch.bfloat16)
        g, b = torch.ones(cols, device=dev), torch.zeros(cols, device=dev)
        t = timed(lambda: ops.layernorm_fwd(x, g, b, 1e-12))
        mb = rows * cols / 1e6
        print("%6d x %4d  MX LayerNorm (bf16 in, bf16 + codes out) %6.1f us (%4.2f TB/s)" % (rows, cols, t, 5 * mb / t))
_native.set_gemm_mode(prev)
