"""Gradient arena + data-parallel wrapper on the GPU with the real two-stream model: gradients written in place by the
native backward kernels must equal the plain (per-call buffer) run, param.grad must alias the arena, the optimizer
step must be unchanged, and DistributedDataParallel (world-size-1 RCCL group, two HIP streams on) must reproduce the
unwrapped gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import helpers
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]


def _model(cfg, sd):
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining
    m = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
    m.load_state_dict(sd)
    return m.to(DEV).train()


def _setup(cfgname="bert_base_2layer_2conect.json", batch=8):
    cfg = synth.load_config(cfgname) if cfgname != "tiny" else synth.tiny_config()
    sd = synth.make_state_dict(cfg, "pretraining")
    x = synth.make_inputs(cfg, batch, 20, 37, with_labels=True) if cfgname != "tiny" else \
        synth.make_inputs(cfg, batch, 9, 8, with_labels=True)
    return cfg, sd, [x[n].to(DEV) for n in NAMES]


def _grads(model):
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def _assert_same(a, b, what, tol=2e-5):
    assert a.keys() == b.keys(), what
    gmax = max(t.abs().max().item() for t in a.values())
    for n in a:
        err = (a[n] - b[n]).abs().max().item()
        assert err <= tol * a[n].abs().max().item() + 1e-6 * gmax, "%s: %s differs by %.3e" % (what, n, err)


@pytest.fixture
def no_dropout():
    import vilbert.vilbert as V
    orig = V._drop_p
    V._drop_p = lambda m: 0.0
    yield
    V._drop_p = orig


def test_arena_gradients_equal_plain_run_and_alias_the_buffer(no_dropout):
    from vilbert import arena as A
    cfg, sd, args = _setup()
    plain = _model(cfg, sd)
    sum(l.sum() for l in plain(*args)).backward()
    want = _grads(plain)

    model = _model(cfg, sd)
    ar = A.GradArena(list(reversed(list(model.parameters()))))
    try:
        sum(l.sum() for l in model(*args)).backward()
        torch.cuda.synchronize()
        got = _grads(model)
        _assert_same(want, got, "arena vs plain")
        n_alias = 0
        for p in model.parameters():
            e = A.lookup(p)
            if p.grad is not None:
                assert p.grad.data_ptr() == ar.views[e[1]].data_ptr(), "gradient was copied instead of written in place"
                n_alias += 1
        assert n_alias > 100
        # tied word embeddings / decoder: ONE parameter, both contributions accumulated in place
        assert model.cls.predictions.decoder.weight is model.bert.embeddings.word_embeddings.weight
        # second pass without zeroing accumulates; zero_grad(set_to_none) starts over (one fill of the arena)
        sum(l.sum() for l in model(*args)).backward()
        torch.cuda.synchronize()
        _assert_same({n: 2 * g for n, g in want.items()}, _grads(model), "accumulated", tol=4e-5)
        model.zero_grad(set_to_none=True)
        sum(l.sum() for l in model(*args)).backward()
        torch.cuda.synchronize()
        _assert_same(want, _grads(model), "after zero_grad")
        # never-used parameters keep grad None and a zero slice
        q = model.bert.encoder.c_layer[0].biOutput.q_dense1.weight
        assert q.grad is None and float(ar.views[A.lookup(q)[1]].abs().max()) == 0.0
    finally:
        ar.release()


def test_adamw_creates_an_arena_and_training_matches_the_plain_run(no_dropout):
    from vilbert import arena as A
    from vilbert.optim import AdamW
    cfg, sd, args = _setup("tiny", 6)

    def run(use_arena):
        model = _model(cfg, sd)
        opt = AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
        if not use_arena:
            opt._arena.release()
        losses = []
        for _ in range(4):
            opt.zero_grad()
            loss = sum(l.sum() for l in model(*args))
            loss.backward()
            opt.step()
            losses.append(loss.item())
        if use_arena:
            assert all(A.lookup(p) is not None for p in model.parameters())
            opt._arena.release()
        return losses, {n: p.detach().clone() for n, p in model.named_parameters()}
    l0, w0 = run(False)
    l1, w1 = run(True)
    assert l0 == pytest.approx(l1, rel=2e-5)
    for n in w0:
        assert torch.allclose(w0[n], w1[n], rtol=1e-4, atol=2e-6), n


def test_ddp_world_size_one_with_two_streams_matches_unwrapped(no_dropout):
    """DistributedDataParallel over an RCCL group of one rank, small buckets (several in-place all-reduces launched
    during backward from both HIP streams): same gradients as the unwrapped model, three steps in a row."""
    import vilbert.vilbert as V
    from vilbert.distributed import DistributedDataParallel as DDP
    cfg, sd, args = _setup()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    prev = V.set_two_streams(True)
    try:
        def run(wrap, **kw):
            m = _model(cfg, sd)
            w = DDP(m, message_size=4 * 1024 * 1024, **kw) if wrap else m
            out = []
            for _ in range(3):
                w.zero_grad()
                sum(l.sum() for l in w(*args)).backward()
                torch.cuda.synchronize()
                out.append(_grads(m))
            if wrap:
                assert len(w._buckets) > 3
                for p in m.parameters():
                    if p.grad is not None:
                        assert p.grad.data_ptr() == w.arena.views[w._where[id(p)][1]].data_ptr()
                w.arena.release()
            return out
        plain, wrapped = run(False), run(True)
        for a, b in zip(plain, wrapped):
            _assert_same(a, b, "DDP vs plain")
        # the two-phase exchange (in-place reduce_scatter_tensor into a shard OF the bucket + all_gather_into_tensor back) on RCCL
        # itself - a group of one rank normally short-cuts to all_reduce - with fp32 and with bf16 buckets (round-4 review)
        real_rs, calls = dist.reduce_scatter_tensor, []
        dist.reduce_scatter_tensor = lambda *a, **k: (calls.append(a[0].numel()), real_rs(*a, **k))[1]
        try:
            direct = run(True, algorithm="direct", direct_at_world_size_one=True)
        finally:
            dist.reduce_scatter_tensor = real_rs
        assert len(calls) > 9, "the direct exchange must have run (%d reduce_scatter launches)" % len(calls)
        for a, b in zip(plain, direct):
            _assert_same(a, b, "DDP direct (reduce_scatter + all_gather at world size 1) vs plain")
        direct16 = run(True, algorithm="direct", direct_at_world_size_one=True, bucket_dtype=torch.bfloat16)
        for a, b in zip(plain, direct16):
            for n in a:
                assert torch.allclose(a[n], b[n], rtol=1e-2, atol=1e-2 * float(a[n].abs().max()) + 1e-12), n
    finally:
        V.set_two_streams(prev)
        dist.destroy_process_group()


def test_weight_gradient_side_streams_give_the_same_gradients(no_dropout):
    """wgrad GEMMs on side streams of the text / image backward streams (autograd_ops._wgrad) vs everything in order:
    same gradients over three passes (tied decoder / word-embedding slice: second writer joins the side streams),
    same optimizer trajectory."""
    import vilbert.vilbert as V
    from vilbert import autograd_ops as A
    from vilbert.optim import AdamW
    cfg, sd, args = _setup()
    prev2 = V.set_two_streams(True)
    try:
        def run(on):
            prev = A.set_wgrad_stream(on)
            try:
                m = _model(cfg, sd)
                opt = AdamW(m.parameters(), lr=1e-3)
                out = []
                for _ in range(3):
                    opt.zero_grad()
                    sum(l.sum() for l in m(*args)).backward()
                    torch.cuda.synchronize()
                    out.append(_grads(m))
                    opt.step()
                return out, {n: p.detach().clone() for n, p in m.named_parameters()}
            finally:
                A.set_wgrad_stream(prev)
        (g_off, w_off), (g_on, w_on) = run(False), run(True)
        assert A._WGRAD["streams"], "the side streams were never used"
        for a, b in zip(g_off, g_on):
            _assert_same(a, b, "wgrad streams on vs off", tol=1e-4)
        _assert_same(w_off, w_on, "weights after 3 steps", tol=1e-4)
    finally:
        V.set_two_streams(prev2)
