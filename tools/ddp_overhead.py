"""Where does the data-parallel machinery spend its time at world size 1? (one GPU)
    python tools/ddp_overhead.py
Times the bench train step (B=256) plain, wrapped with the collectives stubbed out (hooks + bookkeeping only), wrapped
normally (one-rank RCCL all-reduce of ~1 GB in 4-5 buckets), and the stand-alone one-rank all-reduce of the arena."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from vilbert.distributed import DistributedDataParallel as DDP  # noqa: E402
from vilbert.optim import AdamW  # noqa: E402
from vilbert.vilbert import BertConfig  # noqa: E402

cfg = BertConfig.from_json_file(os.path.join(ROOT, "vilbert-multi-task_amd", "config", bench.CONFIG)).to_dict()
x = bench.synthetic_batch(cfg, 256, 36, 37, 7, True)
names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
inp = tuple(x[n].to(dev) for n in names)


def run(wrap, stub):
    model = bench.build_model(cfg, "pretraining", dev).train()
    mib = int(os.environ.get("DDP_MIB", "64"))
    net = DDP(model, message_size=mib * (1 << 20) // 4, delay_allreduce=os.environ.get("DDP_DELAY") == "1") if wrap else model
    if wrap:
        print("buckets:", len(net._buckets), "of", mib, "MiB; delay_allreduce:", net.delay_allreduce)
    opt = AdamW(net.parameters(), lr=1e-4)
    real = dist.all_reduce

    class _Done(object):
        def wait(self):
            return True
    if stub:
        dist.all_reduce = lambda *a, **k: _Done()
    try:
        def step():
            opt.zero_grad(set_to_none=True)
            sum(l.mean() for l in net(*inp)).backward()
            opt.step()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 8
    finally:
        dist.all_reduce = real
        if wrap:
            net.arena.release()
    return ms


plain = run(False, False)
hooks = run(True, True)
full = run(True, False)
buf = torch.zeros(250_000_000, device=dev)
for _ in range(2):
    dist.all_reduce(buf)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dist.all_reduce(buf)
torch.cuda.synchronize()
ar = 1e3 * (time.perf_counter() - t0) / 5
def _time_ar(t, op):
    for _ in range(2):
        dist.all_reduce(t, op=op)
    torch.cuda.synchronize()
    t0_ = time.perf_counter()
    for _ in range(5):
        dist.all_reduce(t, op=op)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0_) / 5


print("stand-alone one-rank all_reduce: fp32 1 GB SUM %.2f ms, AVG %.2f ms; bf16 0.5 GB SUM %.2f ms, AVG %.2f ms" % (
    _time_ar(buf, dist.ReduceOp.SUM), _time_ar(buf, dist.ReduceOp.AVG), _time_ar(buf.bfloat16(), dist.ReduceOp.SUM),
    _time_ar(buf.bfloat16(), dist.ReduceOp.AVG)))
print("plain step %.2f ms | DDP hooks + bookkeeping only %.2f ms (+%.1f %%) | DDP with one-rank RCCL all-reduce %.2f ms "
      "(+%.1f %%) | stand-alone one-rank all-reduce of 1 GB: %.2f ms" % (plain, hooks, 100 * (hooks / plain - 1), full,
                                                                          100 * (full / plain - 1), ar))
dist.destroy_process_group()
