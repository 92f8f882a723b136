"""Micro-benchmark of the fp8 GEMM kernel through the C ABI (HIP events over back-to-back launches).
   python tools/fp8_lab.py [M N K residual act] ...   (default: the forward shapes of the B=512 encoder)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vilbert-multi-task_amd"))
from vilbert import _native as N  # noqa: E402
from vilbert import ops  # noqa: E402

SHAPES = [(18432, 768, 768, 1, None), (18432, 2304, 768, 0, None), (18432, 3072, 768, 0, "gelu"),
          (18432, 768, 3072, 1, None), (18432, 1024, 1024, 1, None), (18432, 1024, 1024, 0, "gelu"),
          (18432, 3072, 1024, 0, None)]


def run(M, Nn, K, res, act, reps=20):
    dev = "cuda:0"
    x = torch.randn(M, K, device=dev)
    w = torch.randn(Nn, K, device=dev) * 0.03
    b = torch.randn(Nn, device=dev)
    r = torch.randn(M, Nn, device=dev) if res else None
    y = torch.empty(M, Nn, device=dev)
    xq, xs = ops.quantize_rows_fp8(x)
    wq, ws = ops.quantize_rows_fp8(w)
    a = N.LinearFp8Args()
    a.A, a.lda, a.a_scale = xq.data_ptr(), K, xs.data_ptr()
    a.W, a.ldw, a.w_scale = wq.data_ptr(), K, ws.data_ptr()
    a.bias = b.data_ptr()
    a.C, a.ldc = y.data_ptr(), Nn
    if r is not None:
        a.residual, a.ldr = r.data_ptr(), Nn
    a.M, a.N, a.K, a.act = M, Nn, K, N.ACT_CODES[act]
    st = N.stream_ptr()
    for _ in range(3):
        N.check(N.lib().vb_linear_fwd_fp8(st, ctypes.byref(a)), "fp8")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        N.lib().vb_linear_fwd_fp8(st, ctypes.byref(a))
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    byts = M * Nn * 4 * (2 if res else 1) + M * K + Nn * K
    # quantiser
    e0.record()
    for _ in range(reps):
        ops.quantize_rows_fp8(x, xq, xs)
    e1.record()
    torch.cuda.synchronize()
    qus = 1e3 * e0.elapsed_time(e1) / reps
    print("M=%6d N=%5d K=%5d res=%d act=%-5s  %8.1f us  %7.1f TF  %5.2f TB/s (min traffic)   quantise A: %6.1f us %5.2f TB/s"
          % (M, Nn, K, res, act, us, 2.0 * M * Nn * K / us / 1e6, byts / us / 1e6, qus, M * K * 5 / qus / 1e6), flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    shapes = SHAPES
    if args:
        shapes = [(int(args[0]), int(args[1]), int(args[2]), int(args[3]), None if args[4] == "none" else args[4])]
    print("VB_FP8_ABL=%s" % os.environ.get("VB_FP8_ABL", "0"))
    for s in shapes:
        run(*s)
