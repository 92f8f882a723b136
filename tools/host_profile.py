"""Where the HOST time of one eager training step goes (the per-GPU batch of BASELINE configs[2], 64 samples, is
bound by the launching thread, not by the GPU).

    python tools/host_profile.py [--batch 64] [--steps 20] [--top 45]

Prints the wall time per step (device synchronised every `steps`), the time the host needs to ENQUEUE a step, and a
cProfile table of the enqueueing thread sorted by own time."""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--config", default="bert_base_6layer_6conect.json")
    args = ap.parse_args()
    import bench
    from vilbert.optim import AdamW
    from vilbert.vilbert import BertConfig
    device = torch.device("cuda:0")
    cfg = BertConfig.from_json_file(os.path.join(ROOT, "vilbert-multi-task_amd", "config", args.config)).to_dict()
    xb = bench.synthetic_batch(cfg, args.batch, 36, 37, 7, True)
    names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask",
             "image_attention_mask", "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
    inp = tuple(xb[n].to(device) for n in names)
    net = bench.build_model(cfg, "pretraining", device).train()
    if os.environ.get("VB_LABEL_GATHER", "auto") == "auto":
        net.label_capacity = "auto"       # sync-free gather of the labelled rows (what bench.py times by default since round 5)
    opt = AdamW(net.parameters(), lr=1e-4, betas=(0.9, 0.98), weight_decay=0.01)

    def step():
        opt.zero_grad(set_to_none=True)
        lm, img, nsp = net(*inp)
        loss = lm.mean() + img.mean() + nsp.mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("batch %d: enqueue %.2f ms/step, wall %.2f ms/step (%.0f samples/s), drain after the last enqueue %.2f ms"
          % (args.batch, 1e3 * (t1 - t0) / args.steps, 1e3 * (t2 - t0) / args.steps,
             args.batch * args.steps / (t2 - t0), 1e3 * (t2 - t1)))
    # forward / backward / optimizer split of the enqueue time
    tf = tb = to = 0.0
    for _ in range(args.steps):
        a = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        lm, img, nsp = net(*inp)
        loss = lm.mean() + img.mean() + nsp.mean()
        b = time.perf_counter()
        loss.backward()
        c = time.perf_counter()
        opt.step()
        d = time.perf_counter()
        tf, tb, to = tf + b - a, tb + c - b, to + d - c
    torch.cuda.synchronize()
    print("enqueue split: forward %.2f ms, backward %.2f ms, optimizer %.2f ms"
          % (1e3 * tf / args.steps, 1e3 * tb / args.steps, 1e3 * to / args.steps))
    # the autograd engine runs backward nodes on its own thread, which cProfile does not see: keep it on this one
    torch.autograd.set_multithreading_enabled(False)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("single-threaded autograd: enqueue %.2f ms/step" % (1e3 * (t1 - t0) / args.steps))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(args.top)
    print("(cProfile over %d steps; its own overhead inflates the totals)" % args.steps)
    print(s.getvalue())


if __name__ == "__main__":
    main()
