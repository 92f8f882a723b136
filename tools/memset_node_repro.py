"""Minimal reproducer behind the round-5 "fp8 chain graph replays wrong losses" issue:
does a hipMemsetAsync / hipMemset2DAsync issued on a CAPTURING stream take effect when the graph is replayed?

    python tools/memset_node_repro.py

No kernel of this repository is involved: torch provides the capture (torch.cuda.graph) and the kernels around the memset,
the memsets go straight to libamdhip64 through ctypes on torch's capture stream - exactly what csrc/gemm.hip did in front of
its split-K launches (`hipMemset2DAsync(dX ...)` then kernels that ADD into dX) until round 6.

Every variant captures   [memset(buf) -> out = buf + 1]   (+ optionally later nodes that dirty the same memory again, as the
caching allocator's block reuse inside a captured step does), dirties `buf` between replays and checks out == 1 after each.
"""
import ctypes
import sys

import torch

dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemset2DAsync.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]


def raw_stream():
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def memset1d(t):
    return hip.hipMemsetAsync(ctypes.c_void_p(t.data_ptr()), 0, t.numel() * t.element_size(), raw_stream())


def memset2d(t, cols):
    """t [rows, ld] fp32: zero the first `cols` columns of every row (pitch = ld * 4)."""
    return hip.hipMemset2DAsync(ctypes.c_void_p(t.data_ptr()), t.stride(0) * 4, 0, cols * 4, t.shape[0], raw_stream())


def run(name, build, replays=5):
    """build() is called under capture and returns (buf_to_dirty_or_None, out, expected_value)."""
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        buf, out, want = build()
    bad = []
    for r in range(replays):
        if buf is not None:
            buf.fill_(1e30)          # eager, between replays: what the memset node has to clear
        g.replay()
        torch.cuda.synchronize()
        lo, hi = float(out.min()), float(out.max())
        if not (lo == want and hi == want):
            bad.append((r, lo, hi))
    print("%-72s %s" % (name, "ok" if not bad else "MEMSET NODE HAD NO EFFECT at replays %s" % bad))
    return not bad


ok = True
static = torch.full((32, 768), 5.0, device=dev)
wide = torch.full((32, 30524), 5.0, device=dev)


def v_static_1d():
    assert memset1d(static) == 0
    return static, static + 1.0, 1.0


def v_static_2d():
    assert memset2d(static, 768) == 0
    return static, static + 1.0, 1.0


def v_static_2d_pitch():
    assert memset2d(wide, 30522) == 0
    return wide, wide[:, :30522] + 1.0, 1.0


def v_pool_2d_reused(pre_kernels=0):
    """The step's pattern: the buffer is allocated INSIDE the capture (graph-private pool), memset, accumulated into, freed, and
    its block is reused by a later allocation of the same capture that leaves large values behind for the next replay."""
    def build():
        x = torch.ones(8, device=dev)
        for _ in range(pre_kernels):
            x = x * 1.0001
        t = torch.empty((32, 768), device=dev)
        assert memset2d(t, 768) == 0
        out = t + 1.0
        ptr = t.data_ptr()
        del t
        u = torch.empty((32, 768), device=dev)     # same size: the allocator hands the freed block back
        u.fill_(1e30)
        build.reused = u.data_ptr() == ptr
        build.keep = (u, x)
        return None, out, 1.0
    return build


ok &= run("static buffer, hipMemsetAsync", v_static_1d)
ok &= run("static buffer, hipMemset2DAsync (pitch == width)", v_static_2d)
ok &= run("static buffer, hipMemset2DAsync (pitch 30524 floats, width 30522)", v_static_2d_pitch)
for n in (0, 64, 1024):
    b = v_pool_2d_reused(n)
    ok &= run("graph-pool buffer, memset2D, block reused later in the capture (%d kernels in front)" % n, b)
    print("    (later allocation reused the block: %s)" % b.reused)
print("RESULT:", "memset nodes behave" if ok else "memset nodes are NOT reliable under graph replay on this runtime")
sys.exit(0)
