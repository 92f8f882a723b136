"""Which ATen ops (fills, adds, copies, ...) does one train_concap step still launch, from where?
   python tools/aten_census.py   -> table of aten ops with shapes + top Python call sites (torch.profiler, one step)."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
import bench  # noqa: E402
from oracle import synth  # noqa: E402
from vilbert.optim import AdamW  # noqa: E402
from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining  # noqa: E402

B = int(os.environ.get("B", "256"))
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


for _ in range(3):
    step()
torch.cuda.synchronize()
# Python-level call sites of torch.zeros in one step
import traceback
_zeros, _calls = torch.zeros, Counter()


def _spy(*a, **k):
    fr = [f for f in traceback.extract_stack()[:-1] if "vilbert" in f.filename][-2:]
    _calls[(str(a[0])[:40], " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr)))] += 1
    return _zeros(*a, **k)


torch.zeros = _spy
step()
torch.zeros = _zeros
torch.cuda.synchronize()
for (shape, where), n in _calls.most_common(12):
    print("torch.zeros x%-3d %-42s %s" % (n, shape, where))
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step()
torch.cuda.synchronize()
ops = Counter()
sites = {}
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    if e.name in ("aten::empty", "aten::empty_like", "aten::empty_strided", "aten::view", "aten::as_strided", "aten::slice",
                  "aten::select", "aten::reshape", "aten::detach", "aten::alias", "aten::_unsafe_view", "aten::unsqueeze",
                  "aten::squeeze", "aten::expand", "aten::transpose", "aten::t", "aten::permute", "aten::narrow",
                  "aten::result_type", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::to",
                  "aten::resolve_conj", "aten::resolve_neg", "aten::unbind", "aten::size", "aten::stride", "aten::is_nonzero",
                  "aten::contiguous", "aten::view_as", "aten::flatten", "aten::unflatten", "aten::chunk", "aten::split"):
        continue
    key = (e.name, str(e.input_shapes)[:70])
    ops[key] += 1
    st = [s for s in (e.stack or []) if "vilbert" in s or "bench" in s or "aten_census" in s]
    sites.setdefault(key, Counter())[st[0][-90:] if st else "?"] += 1
for (name, shp), n in ops.most_common(45):
    top = ", ".join("%s x%d" % (k, v) for k, v in sites[(name, shp)].most_common(2))
    print("%4d  %-28s %-70s %s" % (n, name, shp, top))
