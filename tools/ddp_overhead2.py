"""Finer breakdown of tools/ddp_overhead.py (world size 1, collectives stubbed): which part of the wrapper costs what.
    python tools/ddp_overhead2.py        (VB_GEMM_MODE=bf16 for the bf16 step)
variants: plain | wrapped, hooks replaced by nothing (every bucket 'exchanged' at the end of backward) | wrapped, hooks on,
delay_allreduce=True | wrapped normally (stubbed collectives)."""
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
os.environ.setdefault("MASTER_PORT", "29578")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from vilbert import distributed as D  # noqa: E402
from vilbert.optim import AdamW  # noqa: E402
from vilbert.vilbert import BertConfig  # noqa: E402

cfg = BertConfig.from_json_file(os.path.join(ROOT, "vilbert-multi-task_amd", "config", bench.CONFIG)).to_dict()
B = int(os.environ.get("B", "256"))
x = bench.synthetic_batch(cfg, B, 36, 37, 7, True)
names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
inp = tuple(x[n].to(dev) for n in names)


class _Done(object):
    def wait(self):
        return True


def run(kind):
    model = bench.build_model(cfg, "pretraining", dev).train()
    model.label_capacity = "auto"
    real_ar, real_hook = dist.all_reduce, D.DistributedDataParallel._hook
    if kind == "nohooks":
        D.DistributedDataParallel._hook = lambda self, param: None
    if kind == "v1":          # the pass is entered (so _finalize runs and 'launches' every bucket), nothing else per parameter
        D.DistributedDataParallel._hook = lambda self, param: self._enter_pass()
    if kind == "v2":          # + the bookkeeping, without touching param.grad / the current stream
        def v2(self, param):
            self._enter_pass()
            b, i = self._where[id(param)]
            b.ready.add(i)
        D.DistributedDataParallel._hook = v2
    if kind == "v3":          # + param.grad
        def v3(self, param):
            self._enter_pass()
            b, i = self._where[id(param)]
            g = param.grad
            b.ready.add(i)
        D.DistributedDataParallel._hook = v3
    acc = {"hook": 0.0, "fin": 0.0, "n": 0}
    if kind in ("delay", "normal") and os.environ.get("DDP_TIMERS"):
        h0, f0 = D.DistributedDataParallel._hook, D.DistributedDataParallel._finalize

        def th(self, param):
            t = time.perf_counter()
            h0(self, param)
            acc["hook"] += time.perf_counter() - t
            acc["n"] += 1

        def tf(self):
            t = time.perf_counter()
            f0(self)
            acc["fin"] += time.perf_counter() - t
        D.DistributedDataParallel._hook, D.DistributedDataParallel._finalize = th, tf
    net = model if kind == "plain" else D.DistributedDataParallel(model, delay_allreduce=(kind != "normal"))
    opt = AdamW(net.parameters(), lr=1e-4)
    if kind != "plain":
        dist.all_reduce = lambda *a, **k: _Done()
    try:
        def step():
            opt.zero_grad(set_to_none=True)
            sum(l.mean() for l in net(*inp)).backward()
            opt.step()
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    finally:
        dist.all_reduce, D.DistributedDataParallel._hook = real_ar, real_hook
        if "f0" in dir():
            D.DistributedDataParallel._finalize = f0
        if kind != "plain":
            net.arena.release()
    if acc["n"]:
        print("   host time inside the hooks %.2f ms/step (%d calls/step), inside _finalize %.2f ms/step" % (
            1e3 * acc["hook"] / 14, acc["n"] // 14, 1e3 * acc["fin"] / 14), flush=True)
    return 1e2 * (t2 - t0), 1e2 * (t1 - t0)


for kind in (os.environ.get("KINDS", "plain,nohooks,delay,normal,plain").split(",")):
    wall, enq = run(kind)
    print("%-8s wall %.2f ms/step, host enqueue %.2f ms/step" % (kind, wall, enq), flush=True)
dist.destroy_process_group()
