"""N = 2 data-parallel run of the REAL two-stream model on ONE GPU (round-2 verdict, weak point 2).

The bucket / hook / side-stream-join logic of vilbert/distributed.py had met the real model only at world size 1 and
a world of two ranks only with a toy net on CPU tensors. RCCL refuses two ranks on the same device, so the two
processes of this test share cuda:0 and exchange their arena buckets with the gloo backend (it all-reduces device
tensors through the host): everything on this side of the collective - zero-copy arena buckets, early launch from the
post-accumulate hooks, two producer streams + weight-gradient side streams joined before a bucket is reduced, the
538-tensor parameter set with its never-used q_dense1/2 and the tied word-embedding / MLM-decoder matrix, AdamW on
the averaged slices - is the code an 8-GPU RCCL run executes. Both modes the reference uses: overlapped
(train_concap.py:513) and delay_allreduce=True (train_tasks.py:497).

Per configuration, three optimizer steps at 4 samples per rank; rank 0 then repeats them in ONE process on the
concatenated batch with loss = mean of the two half-batch losses (what averaging the ranks' gradients computes) and
compares every gradient of every step and the final weights; for bert_base_2layer_2conect the first step's gradients
are also compared with autograd through the CPU oracle."""
import os
import socket
import sys
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
PER_RANK, STEPS, LR = 4, 3, 1e-3


def _loss(outs):
    return sum(o.mean() for o in outs)


def _worker(rank, world, port, cfgname, delay, algorithm, q):
    try:
        for p in (os.path.join(ROOT, "vilbert-multi-task_amd"), ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import vilbert.vilbert as V
        from apex.parallel import DistributedDataParallel as DDP    # the import the reference scripts use
        from oracle import synth
        from vilbert import autograd_ops as AO
        from vilbert.optim import AdamW
        from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining
        V._drop_p = lambda m: 0.0
        V.set_two_streams(True)
        AO.set_wgrad_stream(True)
        dev = "cuda:0"
        cfg = synth.load_config(cfgname)
        n_tok, n_reg = (20, 37) if "2layer" in cfgname else (36, 37)
        sd0 = synth.make_state_dict(cfg, "pretraining", seed=21)
        x = synth.make_inputs(cfg, world * PER_RANK, n_tok, n_reg, seed=21, with_labels=True)
        full = [x[n] for n in NAMES]
        mine = [t[rank * PER_RANK:(rank + 1) * PER_RANK].to(dev) for t in full]

        def build(seed_sd):
            m = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
            m.load_state_dict(seed_sd)
            return m.to(dev).train()

        # a different start on rank 1: the wrapper's constructor must broadcast rank 0's weights
        net = build(sd0 if rank == 0 else synth.make_state_dict(cfg, "pretraining", seed=99))
        ddp = DDP(net, delay_allreduce=delay, message_size=8 * 1024 * 1024, algorithm=algorithm)
        assert ddp.algorithm == algorithm
        assert len(ddp._buckets) >= 3
        opt = AdamW(net.parameters(), lr=LR, weight_decay=0.01)
        grads = []
        for _ in range(STEPS):
            opt.zero_grad(set_to_none=True)
            _loss(ddp(*mine)).backward()
            torch.cuda.synchronize()
            grads.append({n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
            assert not any("q_dense" in n for n in grads[-1]), "biOutput.q_dense1/2 are never in the graph"
            for p in net.parameters():      # zero copy: every gradient lives in its bucket
                if p.grad is not None:
                    assert p.grad.data_ptr() == ddp.arena.views[ddp._where[id(p)][1]].data_ptr()
            opt.step()
        assert net.cls.predictions.decoder.weight is net.bert.embeddings.word_embeddings.weight
        torch.cuda.synchronize()
        # both ranks hold the same weights: compare a checksum through the collective itself
        chk = torch.stack([p.detach().double().sum() for p in net.parameters()]).cpu()
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk)
        assert torch.equal(both[0], both[1]), "ranks diverged"
        worst = {"grad": 0.0, "weight": 0.0, "oracle": 0.0}
        if rank == 0:
            ref = build(sd0)
            ropt = AdamW(ref.parameters(), lr=LR, weight_decay=0.01)
            halves = [[t[r * PER_RANK:(r + 1) * PER_RANK].to(dev) for t in full] for r in range(world)]
            for s in range(STEPS):
                ropt.zero_grad(set_to_none=True)
                (sum(_loss(ref(*h)) for h in halves) / world).backward()
                torch.cuda.synchronize()
                want = {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}
                assert want.keys() == grads[s].keys()
                gmax = max(g.abs().max().item() for g in want.values())
                for n, g in want.items():
                    err = (g - grads[s][n]).abs().max().item()
                    bound = 1e-4 * g.abs().max().item() + 1e-6 * gmax
                    assert err <= bound, "step %d %s: %.3e > %.3e" % (s, n, err, bound)
                    worst["grad"] = max(worst["grad"], err / bound)
                ropt.step()
            for (n, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
                err = (p - r).abs().max().item()
                assert err <= 0.05 * LR + 1e-5 * r.abs().max().item(), "weights %s: %.3e" % (n, err)
                worst["weight"] = max(worst["weight"], err)
            if "2layer" in cfgname:
                # tie the averaged gradients of step 1 to the oracle as well (CPU autograd over both halves)
                from oracle import vilbert_oracle as vo
                leaves = {k: v.clone().requires_grad_(True) for k, v in sd0.items() if k != "cls.predictions.decoder.weight"}
                leaves["cls.predictions.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]
                (sum(_loss(vo.pretraining_forward(leaves, cfg, *[t[r * PER_RANK:(r + 1) * PER_RANK] for t in full]))
                     for r in range(world)) / world).backward()
                gmax = max(v.grad.abs().max().item() for v in leaves.values() if v.grad is not None)
                for n, g in grads[0].items():
                    o = leaves[n].grad
                    err = (g.cpu().double() - o.double()).abs().max().item()
                    bound = 2e-4 * o.abs().max().item() + 2e-7 * gmax
                    assert err <= bound, "oracle, %s: %.3e > %.3e" % (n, err, bound)
                    worst["oracle"] = max(worst["oracle"], err / bound)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", worst))
    except Exception:   # noqa: BLE001 - reported to the parent, which fails the test with the traceback
        q.put((rank, "fail", traceback.format_exc()))


# "direct" = reduce_scatter + all_gather on the arena range (round-3 review item 6), "ring" = one all_reduce per bucket
@pytest.mark.parametrize("cfgname,delay,algorithm", [
    ("bert_base_2layer_2conect.json", False, "ring"), ("bert_base_2layer_2conect.json", True, "ring"),
    ("bert_base_6layer_6conect.json", False, "ring"), ("bert_base_6layer_6conect.json", True, "ring"),
    ("bert_base_2layer_2conect.json", False, "direct"), ("bert_base_2layer_2conect.json", True, "direct"),
    ("bert_base_6layer_6conect.json", False, "direct")])
def test_two_ranks_share_one_gpu(cfgname, delay, algorithm):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cfgname, delay, algorithm, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, status, info in sorted(res):
        if status != "ok" and algorithm == "direct" and "gloo" in str(info).lower() and "support" in str(info).lower():
            pytest.skip("this gloo build has no reduce_scatter / all_gather for device tensors (RCCL has; the CPU-tensor "
                        "path is covered by tests/test_distributed_cpu.py)")
        assert status == "ok", "rank %d:\n%s" % (rank, info)


def _worker_bf16(rank, world, port, cfgname, q):
    """The bf16 TRAINING step under the wrapper's defaults of that mode (round-5 review: bf16 exchange buckets + the direct
    reduce-scatter / all-gather pair): gradients of two optimizer steps against one process over both halves, same mode."""
    try:
        for p in (os.path.join(ROOT, "vilbert-multi-task_amd"), ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import vilbert.vilbert as V
        from apex.parallel import DistributedDataParallel as DDP
        from oracle import synth
        from vilbert import _native
        from vilbert.optim import AdamW
        from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining
        V._drop_p = lambda m: 0.0
        _native.set_gemm_mode("bf16")
        dev = "cuda:0"
        cfg = synth.load_config(cfgname)
        sd0 = synth.make_state_dict(cfg, "pretraining", seed=21)
        x = synth.make_inputs(cfg, world * PER_RANK, 36, 37, seed=21, with_labels=True)
        full = [x[n] for n in NAMES]
        mine = [t[rank * PER_RANK:(rank + 1) * PER_RANK].to(dev) for t in full]

        def build():
            m = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
            m.load_state_dict(sd0)
            return m.to(dev).train()
        net = build()
        ddp = DDP(net, message_size=8 * 1024 * 1024)
        assert ddp.algorithm == "direct" and ddp.bucket_dtype is torch.bfloat16, (ddp.algorithm, ddp.bucket_dtype)
        opt = AdamW(net.parameters(), lr=LR, weight_decay=0.01)
        grads = []
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            _loss(ddp(*mine)).backward()
            torch.cuda.synchronize()
            grads.append({n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None})
            opt.step()
        chk = torch.stack([p.detach().double().sum() for p in net.parameters()]).cpu()
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk)
        assert torch.equal(both[0], both[1]), "ranks diverged"
        info = {}
        if rank == 0:
            ref = build()
            ropt = AdamW(ref.parameters(), lr=LR, weight_decay=0.01)
            halves = [[t[r * PER_RANK:(r + 1) * PER_RANK].to(dev) for t in full] for r in range(world)]
            for s in range(2):
                ropt.zero_grad(set_to_none=True)
                (sum(_loss(ref(*h)) for h in halves) / world).backward()
                torch.cuda.synchronize()
                want = {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}
                assert want.keys() == grads[s].keys()
                rel = sorted(float((grads[s][n].double() - g.double()).norm() / g.double().norm().clamp_min(1e-12))
                             for n, g in want.items() if not (".key" in n and n.endswith("bias")))
                info["step%d" % s] = (rel[len(rel) // 2], rel[-1])
                # bf16 buckets round every averaged gradient to 8 bits of mantissa (2^-9 relative per element). The second step
                # also carries the weights' divergence of the first: AdamW's first update is lr * sign(g) wherever v is fresh, so
                # a gradient element whose sign the bucket rounding flipped moves its weight by 2 lr - a sanity bound only there
                bound = (6e-3, 5e-2) if s == 0 else (5e-2, 2.0)
                assert rel[len(rel) // 2] <= bound[0] and rel[-1] <= bound[1], "step %d: median %.2e worst %.2e" % (
                    s, rel[len(rel) // 2], rel[-1])
                ropt.step()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", info))
    except Exception:   # noqa: BLE001
        q.put((rank, "fail", traceback.format_exc()))


def test_two_ranks_share_one_gpu_bf16_step_with_its_default_exchange():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bf16, args=(r, 2, port, "bert_base_2layer_2conect.json", q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, status, info in sorted(res):
        if status != "ok" and "gloo" in str(info).lower() and ("support" in str(info).lower() or "bfloat16" in str(info).lower()):
            pytest.skip("this gloo build cannot exchange bfloat16 device tensors (RCCL can; bf16 buckets on CPU tensors: "
                        "tests/test_distributed_cpu.py)")
        assert status == "ok", "rank %d:\n%s" % (rank, info)
        if rank == 0:
            print("bf16 step under DDP (bf16 buckets, direct): gradient relative L2 vs one process, (median, worst) per step:", info)
