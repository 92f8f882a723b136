"""Which host-side op launches the small FillFunctor<float> kernels of one train_concap step (rocprofv3 sees ~108 per step)?
   python tools/fill_census.py [--batch 64]   -> CPU op (with input shapes) -> number of fill kernels, torch.profiler, one step."""
import argparse
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    import bench
    from vilbert.optim import AdamW
    from vilbert.vilbert import BertConfig
    device = torch.device("cuda:0")
    cfg = BertConfig.from_json_file(os.path.join(ROOT, "vilbert-multi-task_amd", "config", "bert_base_6layer_6conect.json")).to_dict()
    xb = bench.synthetic_batch(cfg, args.batch, 36, 37, 7, True)
    names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask",
             "image_attention_mask", "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
    inp = tuple(xb[n].to(device) for n in names)
    net = bench.build_model(cfg, "pretraining", device).train()
    net.label_capacity = "auto"
    opt = AdamW(net.parameters(), lr=1e-4, betas=(0.9, 0.98), weight_decay=0.01)

    def step():
        opt.zero_grad(set_to_none=True)
        lm, img, nsp = net(*inp)
        (lm.mean() + img.mean() + nsp.mean()).backward()
        opt.step()

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    by_op = Counter()
    total = 0
    for ev in prof.events():
        ks = [k for k in getattr(ev, "kernels", []) if "FillFunctor<float>" in k.name]
        if ks:
            total += len(ks)
            stack = [s for s in (ev.stack or []) if "vilbert" in s or "bench" in s or "tools" in s][:2]
            by_op[(ev.name, str(ev.input_shapes)[:60], " <- ".join(s.split("/")[-1] for s in stack))] += len(ks)
    print("FillFunctor<float> kernels in one step: %d" % total)
    for (name, shapes, stack), n in by_op.most_common(25):
        print("%4d  %-22s %-62s %s" % (n, name, shapes, stack))


if __name__ == "__main__":
    main()
