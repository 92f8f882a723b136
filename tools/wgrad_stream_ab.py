"""In-process A/B of the weight-gradient side streams: alternating blocks of train steps with the side streams on / off
(same model, same process, same stream -> queue mapping).   GPU_MAX_HW_QUEUES=<n> python tools/wgrad_stream_ab.py [batch]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import synth  # noqa: E402
from vilbert import autograd_ops as A  # noqa: E402
from vilbert.optim import AdamW  # noqa: E402
from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = synth.load_config("bert_base_6layer_6conect.json")
dev = torch.device("cuda:0")
model = BertForMultiModalPreTraining(BertConfig.from_dict(cfg)).to(dev).train()
opt = AdamW(model.parameters(), lr=1e-4)
xb = bench.synthetic_batch(cfg, B, 36, 37, 7, True)
names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
inp = tuple(xb[n].to(dev) for n in names)


def step():
    opt.zero_grad(set_to_none=True)
    a, b, c = model(*inp)
    (a.mean() + b.mean() + c.mean()).backward()
    opt.step()


for _ in range(4):
    step()
res = {True: [], False: []}
for rep in range(4):
    for on in (True, False):
        A.set_wgrad_stream(on)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        res[on].append(B * 8 / (time.perf_counter() - t0))
print("GPU_MAX_HW_QUEUES=%s B=%d  side streams on: %s   off: %s" % (
    os.environ.get("GPU_MAX_HW_QUEUES", "default"), B, " ".join("%.0f" % v for v in res[True]),
    " ".join("%.0f" % v for v in res[False])))
