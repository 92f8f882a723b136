"""Debug: a graphed fp8-mode training step after other reduced-precision work in the same process.
    python tools/dbg_graph_nan.py <scenario>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
sys.path.insert(0, ROOT)
import vilbert.vilbert as V  # noqa: E402
from oracle import synth  # noqa: E402
from vilbert import _native, ops, ops16  # noqa: E402
from vilbert.graphed import GraphedTrainStep  # noqa: E402
from vilbert.optim import AdamW  # noqa: E402
from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining  # noqa: E402

DEV = "cuda:0"
NAMES = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
sc = sys.argv[1]
V._drop_p = lambda m: 0.0
cfg = synth.load_config("bert_base_2layer_2conect.json")
sd = synth.make_state_dict(cfg, "pretraining")


def model():
    m = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
    m.load_state_dict(sd)
    return m.to(DEV).train()


def eager_steps(mode, n, shape=(8, 36, 37)):
    _native.set_gemm_mode(mode)
    args = [synth.make_inputs(cfg, *shape, seed=70, with_labels=True)[k].to(DEV) for k in NAMES]
    m = model()
    o = AdamW(m.parameters(), lr=2e-4)
    out = []
    for _ in range(n):
        o.zero_grad()
        loss = sum(l.mean() for l in m(*args))
        loss.backward()
        o.step()
        out.append(round(loss.item(), 4))
    _native.set_gemm_mode("f32")
    return out


def graphed(mode, branches, n=6, shape=(4, 12, 10)):
    _native.set_gemm_mode(mode)
    args = [synth.make_inputs(cfg, *shape, seed=70, with_labels=True)[k].to(DEV) for k in NAMES]
    m = model()
    o = AdamW(m.parameters(), lr=3e-4)
    with GraphedTrainStep(m, o, args, warmup=2, branches=branches) as step:
        got = [round(step(*args).item(), 4) for _ in range(n)]
    _native.set_gemm_mode("f32")
    return got


if sc == "S0":
    print(sc, "fp8 chain alone", graphed("fp8", "chain"))
elif sc == "S1":
    print(sc, "bf16 graphed chain", graphed("bf16", "chain", 4, (8, 36, 37)))
    print(sc, "then fp8 chain", graphed("fp8", "chain"))
elif sc == "S2":
    print(sc, "bf16 graphed chain", graphed("bf16", "chain", 4, (8, 36, 37)))
    print(sc, "then fp8 fork", graphed("fp8", "fork"))
elif sc == "S3":
    print(sc, "bf16 graphed chain", graphed("bf16", "chain", 4, (8, 36, 37)))
    ops.fp8_cache_clear(); ops16.shadow_cache_clear(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    print(sc, "caches cleared, then fp8 chain", graphed("fp8", "chain"))
elif sc == "S4":
    print(sc, "bf16 eager", eager_steps("bf16", 3))
    print(sc, "then fp8 chain", graphed("fp8", "chain"))
elif sc == "S5":
    print(sc, "fp8 eager", eager_steps("fp8", 3))
    print(sc, "then fp8 chain", graphed("fp8", "chain"))
elif sc == "S6":
    print(sc, "f32 eager", eager_steps("f32", 3))
    print(sc, "then fp8 chain", graphed("fp8", "chain"))
elif sc == "S7":
    print(sc, "fp8 eager", eager_steps("fp8", 3))
    print(sc, "then f32 chain", graphed("f32", "chain"))
elif sc == "S8":
    print(sc, "fp8 eager", eager_steps("fp8", 3))
    print(sc, "then fp8 EAGER small", eager_steps("fp8", 6, (4, 12, 10)))
