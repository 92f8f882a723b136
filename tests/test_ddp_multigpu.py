"""Two RCCL ranks on two GPUs (skipped on one-GPU boxes): the zero-copy bucketed all-reduce of vilbert/distributed.py
with the real two-stream model and native kernels. Each rank runs half of the batch; the averaged gradients must
equal one GPU running the whole batch with the summed loss halved (sum over samples is additive, so
(g_rank0 + g_rank1) / 2 == grad(0.5 * sum over all samples))."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
NAMES = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]


def _loss(outs):
    return sum(o.sum() for o in outs)        # per-position losses summed: additive over the batch


def _build(dev):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
    from oracle import synth
    import vilbert.vilbert as V
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining
    V._drop_p = lambda m: 0.0                 # the dropout masks depend on the element index inside the local batch
    V.set_two_streams(True)
    cfg = synth.load_config("bert_base_2layer_2conect.json")
    sd = synth.make_state_dict(cfg, "pretraining")
    x = synth.make_inputs(cfg, 8, 20, 37, with_labels=True)
    m = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
    m.load_state_dict(sd)
    return m.to(dev).train(), [x[n] for n in NAMES]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    model, args = _build(dev)
    from vilbert.distributed import DistributedDataParallel as DDP
    ddp = DDP(model, message_size=4 * 1024 * 1024)
    assert len(ddp._buckets) > 3
    half = [a[rank * 4:(rank + 1) * 4].to(dev) for a in args]
    out = []
    for _ in range(2):                         # two passes: the second one uses the learnt unused-parameter set
        ddp.zero_grad()
        _loss(ddp(*half)).backward()
        torch.cuda.synchronize()
        out.append({n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None})
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_two_gpu_rccl_gradients_match_single_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model, args = _build(torch.device("cuda", 0))
    model.zero_grad()
    (0.5 * _loss(model(*[a.to("cuda:0") for a in args]))).backward()
    want = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    gmax = max(t.abs().max().item() for t in want.values())
    for step in range(2):
        for rank in (0, 1):
            got = res[rank][step]
            assert got.keys() == want.keys()
            for n in want:
                err = (torch.from_numpy(got[n]) - want[n]).abs().max().item()
                assert err <= 2e-5 * want[n].abs().max().item() + 1e-6 * gmax, (step, rank, n, err)
