"""Micro-benchmark of the GEMM entry points on the shapes of the 6L/6C model (B=256 -> M=9216).
    [VB_GEMM_BK=16|32] python tools/gemm_bench.py
Prints one line per shape: kind, M, N, K, microseconds, TFLOP/s (algorithmic 2MNK)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
from vilbert import ops  # noqa: E402

M = 9216
FWD = [(M, 768, 768, 1), (M, 768, 768, 3), (M, 3072, 768, 1), (M, 768, 3072, 1), (M, 1024, 1024, 1),
       (M, 1024, 1024, 3), (M, 1024, 768, 3), (M, 1024, 2048, 1), (M, 30522, 768, 1), (256, 1024, 768, 1)]


def timeit(fn, iters=20):
    if os.environ.get("GEMM_BENCH_QUICK"):
        iters = 2
    for _ in range(3 if not os.environ.get("GEMM_BENCH_QUICK") else 1):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    global FWD
    dev = "cuda:0"
    if os.environ.get("GEMM_BENCH_QUICK"):
        FWD = [(M, 3072, 768, 1), (M, 30522, 768, 1)]
    print("VB_GEMM_HYBRID =", os.environ.get("VB_GEMM_HYBRID", "default(1)"), "VB_GEMM_FLAGS =", os.environ.get("VB_GEMM_FLAGS", "0"))
    tot_f = tot_t = 0.0
    for (m, n, k, nseg) in FWD:
        x = torch.randn(m, k, device=dev)
        ws = [torch.randn(n, k, device=dev) * 0.05 for _ in range(nseg)]
        bs = [torch.randn(n, device=dev) for _ in range(nseg)]
        us = timeit(lambda: ops.linear_fwd(x, ws, bs), 10 if n > 10000 else 20)
        fl = 2.0 * m * n * nseg * k
        print("fwd   M=%5d N=%5d K=%5d nseg=%d  %9.1f us  %6.1f TF" % (m, n, k, nseg, us, fl / us / 1e6))
        if m * n * k > 1e9 and n < 10000:
            dy = torch.randn(m, n * nseg, device=dev)
            us = timeit(lambda: ops.linear_bwd_input(dy, ws, k))
            print("dgrad M=%5d N=%5d K=%5d nseg=%d  %9.1f us  %6.1f TF" % (m, n, k, nseg, us, fl / us / 1e6))
            tot_f, tot_t = tot_f + fl, tot_t + us
            us = timeit(lambda: ops.linear_bwd_weight(dy, x, nseg, n, [True] * nseg))
            print("wgrad M=%5d N=%5d K=%5d nseg=%d  %9.1f us  %6.1f TF" % (m, n, k, nseg, us, fl / us / 1e6))
            tot_f, tot_t = tot_f + fl, tot_t + us
        tot_f, tot_t = tot_f + fl, tot_t + us
    print("aggregate: %.1f TF" % (tot_f / tot_t / 1e6))


if __name__ == "__main__":
    main()
