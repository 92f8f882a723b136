"""Experiment: one inference forward over B samples vs the same B samples as `parts` micro-batches, each on its own
stream (free-running, joined at the end).   python tools/microbatch_ab.py f32|fp8 [B] [parts]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import synth  # noqa: E402
import vilbert.vilbert as V  # noqa: E402
from vilbert import _native  # noqa: E402
from vilbert.vilbert import BertConfig, VILBertForVLTasks  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if mode == "fp8":
    _native.set_gemm_mode("fp8")
cfg = synth.load_config("bert_base_6layer_6conect.json")
dev = torch.device("cuda:0")
model = VILBertForVLTasks(BertConfig.from_dict(cfg), num_labels=1).to(dev).eval()
xb = bench.synthetic_batch(cfg, B, 36, 36, 7, False)
names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "co_attention_mask"]
inp = tuple(xb[n].to(dev) for n in names)
h = B // parts
chunks = [tuple(t[i * h:(i + 1) * h].contiguous() for t in inp) for i in range(parts)]
streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
# one image-side stream per micro-batch stream (the package keeps one per device)
_sides = {}


def _side(device):
    k = torch.cuda.current_stream(device).cuda_stream
    if k not in _sides:
        _sides[k] = torch.cuda.Stream(device=device)
    return _sides[k]


V._side_stream = _side


def whole():
    with torch.no_grad():
        return model(*inp)


def split():
    cur = torch.cuda.current_stream()
    outs = []
    with torch.no_grad():
        for st, c in zip(streams, chunks):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(model(*c))
    for st in streams:
        cur.wait_stream(st)
    return outs


for fn in (whole, split):
    for _ in range(3):
        fn()
torch.cuda.synchronize()
res = {"whole": [], "split": []}
for rep in range(3):
    for name, fn in (("whole", whole), ("split", split)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            fn()
        torch.cuda.synchronize()
        res[name].append(B * 6 / (time.perf_counter() - t0))
print("%s B=%d  one forward: %s   %d micro-batches on %d streams: %s" % (
    mode, B, " ".join("%.0f" % v for v in res["whole"]), parts, parts, " ".join("%.0f" % v for v in res["split"])))
