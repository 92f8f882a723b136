"""Forward at the per-GPU batch of BASELINE configs[4] (128) in the MX mode: eager vs one HIP graph, one vs two encoder streams.
    python tools/mx_b128_ab.py [batch]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

import bench  # noqa: E402
from vilbert import _native, vilbert as V  # noqa: E402
from vilbert.graphed import GraphedForward  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
cfg = json.load(open(os.path.join(ROOT, "vilbert-multi-task_amd", "config", bench.CONFIG)))
xb = bench.synthetic_batch(cfg, B, bench.N_TOK, bench.N_REG, 7, False)
names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask", "co_attention_mask"]
inp = tuple(xb[n].to(dev) for n in names)
net = bench.build_model(cfg, "vltasks", dev).eval()


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def eager():
    with torch.no_grad():
        return net(*inp)


for mode in ("fp8", "mxfp8"):
    _native.set_gemm_mode(mode)
    for two in (True, False):
        V.set_two_streams(two)
        e = timed(eager)
        g = GraphedForward(net, inp)
        gt = timed(lambda: g(*inp))
        print("%-6s B=%d two_streams=%-5s eager %7.3f ms (%7.0f samples/s)   graph %7.3f ms (%7.0f samples/s)" % (
            mode, B, two, 1e3 * e, B / e, 1e3 * gt, B / gt), flush=True)
        del g
_native.set_gemm_mode("f32")
