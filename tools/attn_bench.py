"""Micro-benchmark of the short-sequence attention kernels on the shapes of the train_concap step.

    python tools/attn_bench.py [--batch 256] [--iters 50] [--check]

Shapes (per layer): text self-attention 36x36, 12 heads x 64; image self-attention 37x37, 8 x 128; co-attention
(bi_hidden 1024, 8 x 128) text queries over image keys (36 x 37) and image queries over text keys (37 x 36).
q / k / v are column slices of one fused [rows, 3H] projection buffer and the gradients are written into the column
slices of one fused gradient buffer, as in the model. Prints microseconds per launch and the HBM rate of the
algorithmic bytes (forward: q, k, v read + context written; backward: q, k, v, dO read + dq, dk, dv written).
--check compares against a float64 torch reference."""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))


def ref_attention(q, k, v, mask, heads):
    B, Sq, H = q.shape
    Sk, d = k.shape[1], H // heads
    q4 = q.double().view(B, Sq, heads, d).transpose(1, 2)
    k4 = k.double().view(B, Sk, heads, d).transpose(1, 2)
    v4 = v.double().view(B, Sk, heads, d).transpose(1, 2)
    s = q4 @ k4.transpose(-1, -2) / math.sqrt(d) + mask.double().view(B, 1, 1, Sk)
    return (torch.softmax(s, -1) @ v4).transpose(1, 2).reshape(B, Sq, H)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--drop", type=float, default=0.1)
    ap.add_argument("--lib", default=None, help="another build of libvilbert_hip.so to A/B against")
    args = ap.parse_args()
    if args.lib:
        from vilbert import _native
        _native.LIB_PATH = os.path.abspath(args.lib)
        print("library:", _native.LIB_PATH)
    from vilbert import ops
    dev = torch.device("cuda:0")
    B = args.batch
    cases = [("text self 36x36 12x64", 36, 36, 768, 12), ("image self 37x37 8x128", 37, 37, 1024, 8),
             ("co text->image 36x37 8x128", 36, 37, 1024, 8), ("co image->text 37x36 8x128", 37, 36, 1024, 8)]
    torch.manual_seed(5)
    for name, Sq, Sk, H, heads in cases:
        same = Sq == Sk
        if same:      # fused [q | k | v] buffer
            qkv = torch.randn(B, Sq, 3 * H, device=dev)
            q, k, v = qkv[:, :, :H], qkv[:, :, H:2 * H], qkv[:, :, 2 * H:]
            dqkv = torch.empty_like(qkv)
            dq, dk, dv = dqkv[:, :, :H], dqkv[:, :, H:2 * H], dqkv[:, :, 2 * H:]
        else:         # co-attention: q from one stream, fused [k | v] from the other
            q = torch.randn(B, Sq, H, device=dev)
            kv = torch.randn(B, Sk, 2 * H, device=dev)
            k, v = kv[:, :, :H], kv[:, :, H:]
            dq, dkv = torch.empty_like(q), torch.empty_like(kv)
            dk, dv = dkv[:, :, :H], dkv[:, :, H:]
        mask = torch.zeros(B, Sk, device=dev)
        mask[:, Sk - 3:] = -10000.0
        d_out = torch.randn(B, Sq, H, device=dev)
        if args.check:
            out, _, lse = ops.attention_fwd(q, k, v, mask, heads, want_lse=True)
            q64, k64, v64 = (t.detach().double().requires_grad_(True) for t in (q, k, v))
            want = ref_attention(q64, k64, v64, mask, heads)
            want.backward(d_out.double())
            ops.attention_bwd(d_out, q, k, v, mask, heads, lse, dq, dk, dv)
            errs = [((out.double() - want).abs().max() / want.abs().max()).item()]
            for got, w in ((dq, q64.grad), (dk, k64.grad), (dv, v64.grad)):
                errs.append(((got.double() - w).abs().max() / w.abs().max()).item())
            print("%-30s check: max rel err out %.1e dq %.1e dk %.1e dv %.1e" % ((name,) + tuple(errs)))
        _, _, lse = ops.attention_fwd(q, k, v, mask, heads, want_lse=True, drop_p=args.drop, seed=11)

        def timed(fn):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / args.iters
        t_f = timed(lambda: ops.attention_fwd(q, k, v, mask, heads, want_lse=True, drop_p=args.drop, seed=11))
        t_b = timed(lambda: ops.attention_bwd(d_out, q, k, v, mask, heads, lse, dq, dk, dv, drop_p=args.drop, seed=11))
        by_f = 4.0 * B * (2 * Sq + 2 * Sk) * H
        by_b = 4.0 * B * (3 * Sq + 4 * Sk) * H
        print("%-30s fwd %7.1f us %5.2f TB/s | bwd %7.1f us %5.2f TB/s" %
              (name, t_f, by_f / t_f / 1e6, t_b, by_b / t_b / 1e6))


if __name__ == "__main__":
    main()
