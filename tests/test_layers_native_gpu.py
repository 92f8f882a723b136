"""Whole-layer launcher (csrc/layers.hip vb_layer_fwd / vb_layer_bwd, vilbert/layers.py; round 6) against the per-op path.

The launcher enqueues the SAME kernels in the SAME order as the per-op autograd nodes (vilbert/autograd_ops.py) and draws the
dropout seeds in the same order, so in fp32 the two paths must agree BIT FOR BIT - outputs, losses and every parameter
gradient, dropout on - except for the embedding-table gradients, whose scatter adds with fp32 atomics in either path. In the
bf16 mode the layer node adds the skip-connection gradient in the epilogue of the q|k|v input-gradient GEMM (one rounding)
where autograd adds two bf16 tensors (two roundings): agreement to bf16 rounding instead.
The per-op path itself is pinned to the oracle by the rest of the suite (which now runs through the launcher by default:
every model-level parity test is also a test of this file's subject).
Reference: /root/reference/vilbert/vilbert.py:527-533 (BertLayer), 688-694 (BertImageLayer), 871-900 (BertConnectionLayer).
"""
import itertools

import pytest
import torch

import helpers
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]


def _train_step(cfg, sd, args, native, accumulate=1, with_arena=True):
    """loss + gradients of `accumulate` backward passes of a fresh model, dropout ON, seeds restarted."""
    import vilbert.autograd_ops as AO
    from vilbert import layers
    from vilbert.optim import AdamW
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining
    prev = layers.set_native(native)
    try:
        m = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        opt = AdamW(m.parameters(), lr=1e-4) if with_arena else None
        AO._seed_counter = itertools.count(1)
        calls0 = layers.native_calls()
        losses = []
        for _ in range(accumulate):
            out = m(*args)
            loss = sum(l.mean() for l in out)
            loss.backward()
            losses.append([float(l.mean()) for l in out])
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        return losses, grads, layers.native_calls() - calls0, opt
    finally:
        layers.set_native(prev)


ATOMIC = ("word_embeddings", "position_embeddings", "token_type_embeddings", "task_embeddings")


@pytest.mark.parametrize("shape,accumulate,arena", [((3, 9, 7), 1, True), ((4, 36, 37), 1, True), ((2, 12, 10), 2, True),
                                                    ((2, 5, 6), 1, False)])
def test_fp32_layer_launcher_is_bit_identical_to_the_per_op_path_dropout_on(shape, accumulate, arena):
    from vilbert import _native
    cfg = synth.load_config("bert_base_2layer_2conect.json")
    sd = synth.make_state_dict(cfg, "pretraining")
    args = [synth.make_inputs(cfg, *shape, seed=3, with_labels=True)[n].to(DEV) for n in NAMES]
    prev = _native.set_gemm_mode("f32")
    try:
        l_op, g_op, n_op, _o1 = _train_step(cfg, sd, args, native=False, accumulate=accumulate, with_arena=arena)
        l_nat, g_nat, n_nat, _o2 = _train_step(cfg, sd, args, native=True, accumulate=accumulate, with_arena=arena)
    finally:
        _native.set_gemm_mode(prev)
    assert n_op == 0
    # 12 text + 2 image layers: one forward + one backward call each; 2 connection layers: 3 + 3
    assert n_nat == accumulate * (14 * 2 + 2 * 6), n_nat
    assert l_nat == l_op, (l_nat, l_op)
    assert set(g_nat) == set(g_op)
    for n, g in g_op.items():
        if any(k in n for k in ATOMIC) or n.startswith("cls.predictions.decoder"):
            assert torch.allclose(g_nat[n], g, rtol=1e-5, atol=1e-7 * float(g.abs().max()) + 1e-12), n
        else:
            assert torch.equal(g_nat[n], g), "%s: max diff %.3e" % (n, float((g_nat[n] - g).abs().max()))


def test_fp32_layer_launcher_inference_outputs_are_bit_identical():
    from vilbert import _native, layers
    from vilbert.vilbert import BertConfig, VILBertForVLTasks
    cfg = synth.load_config("bert_base_2layer_2conect.json")
    sd = synth.make_state_dict(cfg, "vltasks")
    x = synth.make_inputs(cfg, 5, 20, 37, seed=9, ragged=True)
    args = helpers.to_device((x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
                              x["image_attention_mask"], x["co_attention_mask"]), DEV)
    m = VILBertForVLTasks(BertConfig.from_dict(cfg), num_labels=1)
    m.load_state_dict(sd)
    m = m.eval().to(DEV)
    prev_mode = _native.set_gemm_mode("f32")
    try:
        with torch.no_grad():
            prev = layers.set_native(False)
            want = m(*args)[:9]
            layers.set_native(True)
            c0 = layers.native_calls()
            got = m(*args)[:9]
            assert layers.native_calls() - c0 == 14 + 2 * 3
            layers.set_native(prev)
    finally:
        _native.set_gemm_mode(prev_mode)
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_bf16_layer_launcher_matches_the_per_op_path_to_bf16_rounding():
    from vilbert import _native
    cfg = synth.load_config("bert_base_2layer_2conect.json")
    sd = synth.make_state_dict(cfg, "pretraining")
    args = [synth.make_inputs(cfg, 4, 36, 37, seed=5, with_labels=True)[n].to(DEV) for n in NAMES]
    prev = _native.set_gemm_mode("bf16")
    try:
        l_op, g_op, n_op, _o1 = _train_step(cfg, sd, args, native=False)
        l_nat, g_nat, n_nat, _o2 = _train_step(cfg, sd, args, native=True)
    finally:
        _native.set_gemm_mode(prev)
    assert n_op == 0 and n_nat == 14 * 2 + 2 * 6, (n_op, n_nat)
    for a, b in zip(l_nat[0], l_op[0]):
        assert abs(a - b) <= 2e-3 * abs(b), (l_nat, l_op)
    # (a key bias shifts every score of a query row alike: its gradient is zero in exact arithmetic and rounding noise in
    # either path - compared on the scale of the query bias gradient next to it instead of its own)
    def scale(n, g):
        ref = g_op[n.replace("key", "query")] if ".key" in n and n.endswith("bias") else g
        return ref.double().norm().clamp_min(1e-12)
    named = sorted((float((g_nat[n].double() - g.double()).norm() / scale(n, g)), n) for n, g in g_op.items())
    rel = [r for r, _n in named]
    print("bf16 layer launcher vs per-op path, gradient relative L2: median %.2e, worst %.2e" % (rel[len(rel) // 2], rel[-1]))
    print("  worst five: " + ", ".join("%s %.2e" % (n, r) for r, n in named[-5:]))
    assert rel[len(rel) // 2] <= 1e-2 and rel[-1] <= 6e-2, (rel[len(rel) // 2], rel[-1])


def test_layer_launcher_steps_aside_for_what_it_does_not_serve():
    """Attention maps (visualization), a partially frozen layer and the fp8 inference mode go through the per-op path; a
    frozen PREFIX (fixed_t_layer / fixed_v_layer, reference vilbert.py:968-995) runs through the launcher as inference calls."""
    from vilbert import _native, layers
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining, VILBertForVLTasks
    cfg = synth.load_config("bert_base_2layer_2conect.json")
    x = synth.make_inputs(cfg, 2, 9, 7, seed=1, with_labels=True)
    args = [x[n].to(DEV) for n in NAMES]
    vis = dict(cfg, visualization=True)
    m = VILBertForVLTasks(BertConfig.from_dict(vis), num_labels=1).eval().to(DEV)
    c0 = layers.native_calls()
    with torch.no_grad():
        out = m(*args[:6], output_all_attention_masks=True)
    assert layers.native_calls() == c0 and out[-1][0], "attention maps need the per-op path"
    m = BertForMultiModalPreTraining(BertConfig.from_dict(cfg)).to(DEV).train()
    m.bert.encoder.layer[1].output.dense.weight.requires_grad_(False)           # one frozen weight inside a trainable layer
    c0 = layers.native_calls()
    sum(l.mean() for l in m(*args)).backward()
    n_frozen = layers.native_calls() - c0
    assert n_frozen == (13 * 2 + 2 * 6), n_frozen                                # that layer alone went op by op
    assert m.bert.encoder.layer[1].output.dense.weight.grad is None
    assert m.bert.encoder.layer[1].output.dense.bias.grad is not None
    prev = _native.set_gemm_mode("fp8")
    try:
        m.eval()
        c0 = layers.native_calls()
        with torch.no_grad():
            m(*args[:6])
        assert layers.native_calls() == c0
    finally:
        _native.set_gemm_mode(prev)


def test_replaced_submodules_parameters_and_moved_storage_are_noticed_by_the_plans():
    """The whole-layer nodes keep per-module plans (parameter lists, static argument structs). They must follow what a user
    does to the model between two calls: a sub-module replaced (adapter-style), a Parameter object replaced, the model moved
    to new storage (.cpu().to(dev)) - each time the launcher's outputs equal the per-op path's on the CURRENT weights (and
    differ from the outputs before the edit)."""
    from vilbert import _native, layers
    from vilbert.vilbert import BertConfig, VILBertForVLTasks
    cfg = synth.load_config("bert_base_2layer_2conect.json")
    sd = synth.make_state_dict(cfg, "vltasks")
    x = synth.make_inputs(cfg, 3, 12, 9, seed=4)
    args = helpers.to_device((x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
                              x["image_attention_mask"], x["co_attention_mask"]), DEV)
    m = VILBertForVLTasks(BertConfig.from_dict(cfg), num_labels=1)
    m.load_state_dict(sd)
    m = m.eval().to(DEV)
    prev_mode = _native.set_gemm_mode("f32")

    def both():
        with torch.no_grad():
            prev = layers.set_native(True)
            c0 = layers.native_calls()
            got = [o.clone() for o in m(*args)[:9]]
            assert layers.native_calls() > c0
            layers.set_native(False)
            want = m(*args)[:9]
            layers.set_native(prev)
        for g, w in zip(got, want):
            assert torch.equal(g, w)
        return got
    try:
        base = both()
        g = torch.Generator().manual_seed(3)
        enc = m.bert.encoder
        old = enc.layer[3].attention.self.key
        new = torch.nn.Linear(old.in_features, old.out_features)
        new.weight.data.copy_(torch.randn(new.weight.shape, generator=g) * 0.05)
        enc.layer[3].attention.self.key = new.to(DEV)                                  # a replaced sub-module
        a = both()
        assert not torch.equal(a[0], base[0])
        bi = enc.c_layer[1].biattention
        bi.value2.bias = torch.nn.Parameter((torch.randn(bi.value2.bias.shape, generator=g) * 0.5).to(DEV))   # a replaced Parameter
        b = both()
        assert not torch.equal(b[0], a[0])
        with torch.no_grad():
            enc.v_layer[0].output.dense.weight.mul_(1.5)                               # an in-place edit (same object, same storage)
        c = both()
        assert not torch.equal(c[0], b[0])
        m.cpu()
        m.to(DEV)                                                                      # same objects, new addresses
        d = both()
        for u, v in zip(c, d):
            assert torch.equal(u, v)
    finally:
        _native.set_gemm_mode(prev_mode)
