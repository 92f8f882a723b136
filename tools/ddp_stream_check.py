"""Single-GPU check of DistributedDataParallel (world-size-1 RCCL group) with the two-stream encoder:
its gradients must equal a plain run's up to the atomic-order noise measured between two plain runs."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from oracle import synth  # noqa: E402
import vilbert.vilbert as V  # noqa: E402
from vilbert.distributed import DistributedDataParallel as DDP  # noqa: E402
from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining  # noqa: E402

V._drop_p = lambda m: 0.0
cfg = synth.load_config("bert_base_2layer_2conect.json")
sd = synth.make_state_dict(cfg, "pretraining")
x = synth.make_inputs(cfg, 8, 20, 37, with_labels=True)
names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
args = [x[n].cuda() for n in names]


def grads(wrap):
    m = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
    m.load_state_dict(sd)
    m = m.cuda().train()
    w = DDP(m, message_size=4 * 1024 * 1024) if wrap else m
    out = []
    for _ in range(3):
        w.zero_grad()
        sum(l.sum() for l in w(*args)).backward()
        torch.cuda.synchronize()
        out.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    return out


def worst(a, b):
    w = (0.0, "")
    for ga, gb in zip(a, b):
        assert ga.keys() == gb.keys()
        gmax = max(t.abs().max().item() for t in ga.values())
        for n in ga:
            d = (ga[n] - gb[n]).abs().max().item() / (ga[n].abs().max().item() + 1e-4 * gmax)
            if d > w[0]:
                w = (d, n)
    return w


p1, p2, d1 = grads(False), grads(False), grads(True)
print("plain vs plain      : worst relative grad diff %.3e at %s" % worst(p1, p2))
print("DDP+streams vs plain: worst relative grad diff %.3e at %s" % worst(p1, d1))
noise, real = worst(p1, p2)[0], worst(p1, d1)[0]
assert real <= max(10 * noise, 1e-4), "DDP gradients differ beyond the run-to-run noise"
print("OK")
dist.destroy_process_group()
