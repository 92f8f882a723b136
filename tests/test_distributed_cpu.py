"""N > 1 data-parallel path on CPU: world_size-2 gloo processes exercise the bucketed, overlapped
gradient all-reduce (vilbert/distributed.py) - averaging, unused parameters, tied weights, delayed mode,
set_to_none zero_grad, and equivalence with a single-process run over the concatenated batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.emb = nn.Embedding(11, 8)
        self.a = nn.Linear(8, 16)
        self.unused = nn.Linear(16, 16)        # like biOutput.q_dense1/2: never in the graph
        self.b = nn.Linear(16, 8)
        self.head_x = nn.Linear(8, 3)          # task heads: only one of them gets a gradient per step
        self.head_y = nn.Linear(8, 5)
        self.dec = nn.Linear(8, 11, bias=False)
        self.dec.weight = self.emb.weight      # tied like cls.predictions.decoder / word_embeddings

    def forward(self, ids, task):
        h = self.b(torch.relu(self.a(self.emb(ids))))
        out = self.head_x(h) if task == 0 else self.head_y(h)
        return out.pow(2).mean() + self.dec(h).pow(2).mean()


def _worker(rank, world, port, delay, q):
    sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from apex.parallel import DistributedDataParallel as DDP  # the import the reference scripts use
    torch.manual_seed(100 + rank)           # different init per rank: the constructor must broadcast
    net = Net()
    ddp = DDP(net, delay_allreduce=delay, message_size=200)   # tiny buckets -> several collectives
    assert hasattr(ddp, "module") and len(ddp._buckets) > 2
    g = torch.Generator().manual_seed(7)
    data = torch.randint(0, 11, (4, 6, 5), generator=g)       # [step, global batch, seq]
    grads = []
    for step in range(4):
        ids = data[step][rank * 3:(rank + 1) * 3]
        task = step % 2 if delay else 0     # the unused set may change per step only in delayed mode
        ddp.zero_grad()                      # set_to_none=True
        ddp(ids, task).backward()
        assert net.unused.weight.grad is None
        grads.append({n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
        with torch.no_grad():
            for p in net.parameters():
                if p.grad is not None:
                    p -= 0.1 * p.grad
    # numpy arrays pickle by value (torch tensors would travel as shared-memory handles that die with us)
    q.put((rank, [{n: t.numpy() for n, t in g.items()} for g in grads],
           {n: p.detach().numpy().copy() for n, p in net.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("delay", [False, True])
def test_two_rank_gradient_average_matches_single_process(delay):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, delay, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tt = lambda d: {n: torch.from_numpy(a) for n, a in d.items()}
    (_, g0, w0), (_, g1, w1) = [(r, [tt(g) for g in gs], tt(w)) for r, gs, w in res]
    # both ranks hold identical averaged gradients and identical weights after 4 steps
    for a, b in zip(g0, g1):
        assert a.keys() == b.keys()
        for n in a:
            assert torch.allclose(a[n], b[n], atol=1e-7), n
    for n in w0:
        assert torch.equal(w0[n], w1[n]), n
    # single-process reference: rank 0's initial weights (broadcast), mean of the two half-batch losses
    torch.manual_seed(100)
    net = Net()
    data = torch.randint(0, 11, (4, 6, 5), generator=torch.Generator().manual_seed(7))
    for step in range(4):
        task = step % 2 if delay else 0
        net.zero_grad()
        loss = 0.5 * (net(data[step][:3], task) + net(data[step][3:], task))
        loss.backward()
        for n, p in net.named_parameters():
            if p.grad is not None:
                assert torch.allclose(g0[step][n], p.grad, atol=1e-6), (step, n)
        with torch.no_grad():
            for p in net.parameters():
                if p.grad is not None:
                    p -= 0.1 * p.grad
    for n, p in net.named_parameters():
        assert torch.allclose(w0[n], p, atol=1e-5), n


# ---- exchange algorithms (round-3 review item 6): "direct" = reduce_scatter + all_gather vs "ring" = all_reduce ---------

def _algo_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vilbert.distributed import DistributedDataParallel as DDP
    out = {}
    data = torch.randint(0, 11, (3, 2 * world, 5), generator=torch.Generator().manual_seed(9))
    for algo, dtype, delay in (("ring", None, False), ("direct", None, False), ("direct", None, True),
                               ("ring", torch.bfloat16, False), ("direct", torch.bfloat16, False)):
        torch.manual_seed(5)
        net = Net()
        ddp = DDP(net, delay_allreduce=delay, message_size=150, algorithm=algo, bucket_dtype=dtype)
        assert ddp.algorithm == algo and len(ddp._buckets) > 2
        assert all(b.flat.numel() % world == 0 for b in ddp._buckets)            # whole shards
        assert all(v.data_ptr() % 16 == 0 for v in ddp.arena.views)             # kernels' alignment is kept
        steps = []
        for step in range(3):
            ddp.zero_grad()
            ddp(data[step][rank * 2:(rank + 1) * 2], 0).backward()
            steps.append({n: p.grad.clone().numpy() for n, p in net.named_parameters() if p.grad is not None})
            with torch.no_grad():
                for p in net.parameters():
                    if p.grad is not None:
                        p -= 0.1 * p.grad
        out[(algo, str(dtype), delay)] = steps
        ddp.arena.release()
    with pytest.raises(ValueError):
        DDP(Net(), algorithm="tree")
    # defaults (round 6): fp32 training = ring all-reduce of fp32 buckets; a model in the bf16 mode (`model.half()` before the
    # wrapper, the reference's order train_concap.py:504-513, or the process-wide mode) = direct exchange of bf16 buckets;
    # explicit arguments win
    plain = DDP(Net())
    assert plain.algorithm == "ring" and plain.bucket_dtype is None
    plain.arena.release()
    half = Net()
    half._vb_bf16 = True
    d = DDP(half)
    assert d.algorithm == "direct" and d.bucket_dtype == torch.bfloat16
    d.arena.release()
    d = DDP(half, algorithm="ring", bucket_dtype=torch.float32)
    assert d.algorithm == "ring" and d.bucket_dtype is None
    d.arena.release()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_direct_reduce_scatter_all_gather_equals_all_reduce(world):
    """Same buckets, same gradients: the two-phase exchange must give every rank the SAME averaged gradients as the
    all-reduce. fp32 addition is commutative, so at world 2 the two are bit-identical; at world 4 the association order
    of gloo's ring all-reduce and of its reduce-scatter may differ, hence a 1-ulp-scale bound there. bf16 buckets: both
    algorithms within bf16 rounding of the fp32 result, and identical across ranks."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_algo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ring = res[0][("ring", "None", False)]
    for key in (("direct", "None", False), ("direct", "None", True)):
        for r in range(world):
            for step, want in enumerate(ring):
                got = res[r][key][step]
                assert got.keys() == want.keys()
                for n in want:
                    if world == 2:
                        assert (got[n] == want[n]).all(), (key, r, step, n)
                    else:
                        assert abs(got[n] - want[n]).max() <= 1e-6 * max(1e-3, abs(want[n]).max()), (key, r, step, n)
    # every rank holds the same values (what data-parallel training needs), whatever the algorithm
    for key in res[0]:
        for r in range(1, world):
            for a, b in zip(res[0][key], res[r][key]):
                for n in a:
                    assert (a[n] == b[n]).all(), (key, r, n)
    for key in (("ring", "torch.bfloat16", False), ("direct", "torch.bfloat16", False)):
        for step, want in enumerate(ring):
            for n in want:
                # steps > 0 start from weights that already differ by the rounding of step 0: compare step 0 tightly
                if step == 0:
                    assert abs(res[0][key][0][n] - want[n]).max() <= 2 ** -7 * max(1e-6, abs(want[n]).max()), (key, n)
