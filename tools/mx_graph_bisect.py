"""Why does the MX forward graph at batch 128 replay at 12.6 k samples/s inside bench.py's long leg sequence and at 34 k in a
fresh process (round-4 review, weak 3)? The same GraphedForward (fork and chain forms) is rebuilt and timed after each
thing the default bench line does before that leg:
    0 fresh process                      3 after an fp8-mode graph was captured and destroyed
    1 after N extra torch streams ran    4 after a training step (weight-gradient side streams, arena, AdamW)
    2 after RCCL initialised (world 1)   5 after a whole-step training graph was captured and destroyed
python tools/mx_graph_bisect.py [batch] > profiles/r05_mx_graph_bisect.txt"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

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


def measure(stage):
    _native.set_gemm_mode("mxfp8")
    V.set_two_streams(True)
    e2 = timed(eager)
    V.set_two_streams(False)
    e1 = timed(eager)
    V.set_two_streams(True)
    row = "%-58s eager two streams %6.0f, one stream %6.0f |" % (stage, B / e2, B / e1)
    for form in ("fork", "chain", "auto"):
        g = GraphedForward(net, inp, branches=form, fork_attempts=1 if form == "fork" else 3)
        t = timed(lambda: g(*inp))
        row += " graph %s %6.0f" % (form if form != "auto" else "auto->" + g.branches, B / t)
        del g
    print(row + "  samples/s", flush=True)
    _native.set_gemm_mode("f32")


measure("0 fresh process")

streams = [torch.cuda.Stream(device=dev) for _ in range(12)]
a = torch.randn(2048, 2048, device=dev)
for s in streams:
    with torch.cuda.stream(s):
        (a @ a).sum()
torch.cuda.synchronize()
measure("1 after 12 extra streams ran work (still alive)")

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group(backend="nccl", device_id=dev, rank=0, world_size=1)
t = torch.ones(1 << 20, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
measure("2 after RCCL initialised + one all_reduce (world 1)")

_native.set_gemm_mode("fp8")
g8 = GraphedForward(net, inp, branches="fork")
g8(*inp)
torch.cuda.synchronize()
del g8
measure("3 after an fp8 fork graph was captured, replayed, destroyed")

from vilbert.optim import AdamW  # noqa: E402
tb = bench.synthetic_batch(cfg, 64, bench.N_TOK, bench.N_REG + 1, 7, True)
tnames = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
          "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
tinp = tuple(tb[n].to(dev) for n in tnames)
tnet = bench.build_model(cfg, "pretraining", dev).train()
opt = AdamW(tnet.parameters(), lr=1e-5)
for _ in range(3):
    opt.zero_grad()
    sum(l.mean() for l in tnet(*tinp)).backward()
    opt.step()
torch.cuda.synchronize()
measure("4 after 3 eager training steps (B=64, wgrad side streams)")

from vilbert.graphed import GraphedTrainStep  # noqa: E402
try:
    with GraphedTrainStep(tnet, opt, tinp) as step:
        step(*tinp)
        torch.cuda.synchronize()
    measure("5 after a whole-step training graph was captured + destroyed")
except Exception as e:  # noqa: BLE001 - the bisect must print what it has
    print("5 skipped: %r" % (e,))
del tnet, opt
torch.cuda.empty_cache()
measure("6 after the training model was freed")

# the same fork graph instantiated eight times in a row: the replay rate is a property of the INSTANCE
_native.set_gemm_mode("mxfp8")
rates = []
for i in range(8):
    g = GraphedForward(net, inp, branches="fork", fork_attempts=1)
    rates.append(B / timed(lambda: g(*inp), 20))
    del g
print("7 eight consecutive instances of the fork graph: " + " ".join("%.0f" % r for r in rates) + " samples/s", flush=True)
g = GraphedForward(net, inp)
print("8 auto: kept %s, trials %s" % (g.branches, {k: round(v, 3) for k, v in g.trial_ms.items()}), flush=True)
_native.set_gemm_mode("f32")
