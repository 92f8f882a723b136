"""SURVEY.md 7.3-7(c) loss-curve sanity (round-3 review, weak point 2 / harness item 8): 200 optimizer steps of the tiny
two-stream pre-training model on a fixed set of 8 batches, native AdamW, three runs -

  oracle     CPU, oracle/vilbert_oracle.py autograd + oracle/adamw_oracle.py, no dropout (the oracle has none)
  hip_off    HIP model, dropout probabilities forced to 0      -> must TRACK the oracle curve (same arithmetic, fp32)
  hip_on     HIP model as shipped (train mode, dropout 0.1)    -> must show the dropout signature against it:
             the training loss decreases like the oracle's, stays above the no-dropout curve late in training (the masks
             regularise: the no-dropout runs memorise the 8 batches), and evaluated WITHOUT dropout on the same batches the
             weights it learnt are good - i.e. the masks are drawn afresh every step, scaled by 1 / (1 - p), and absent in
             eval mode. A mask that never changed, a missing rescale or a mask applied in eval would each break one of
             the three bounds.
"""
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
# LR: at 2e-3 this problem is chaotic (a 1e-6 relative perturbation of the initial weights grows to 30 % of the loss by
# step 130 in the CPU oracle itself); at 5e-4 a 1e-4 perturbation stays below 0.1 % of the loss over all 200 steps, so the
# two implementations can be held to the same trajectory
STEPS, NB, LR = 200, 8, 5e-4


def _window(xs, lo, hi):
    return sum(xs[lo:hi]) / (hi - lo)


def test_two_hundred_steps_with_and_without_dropout_against_the_oracle():
    import vilbert.vilbert as V
    from vilbert.optim import AdamW
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining

    cfg = synth.tiny_config()
    sd = synth.make_state_dict(cfg, "pretraining", seed=5)
    batches = [[synth.make_inputs(cfg, 8, 9, 8, seed=50 + i, with_labels=True)[n] for n in NAMES] for i in range(NB)]

    # ---- oracle -------------------------------------------------------------------------------------------------------
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "cls.predictions.decoder.weight"}
    leaves["cls.predictions.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]
    params = {id(v): v for v in leaves.values()}.values()
    state = {id(v): (torch.zeros_like(v), torch.zeros_like(v)) for v in params}
    oracle = []
    for step in range(1, STEPS + 1):
        for v in params:
            v.grad = None
        loss = sum(l.mean() for l in vo.pretraining_forward(leaves, cfg, *batches[(step - 1) % NB]))
        loss.backward()
        oracle.append(loss.item())
        with torch.no_grad():
            for v in params:
                if v.grad is not None:
                    m, s = state[id(v)]
                    ao.adamw_step(v, v.grad, m, s, step, LR, (0.9, 0.999), 1e-6, 0.0, True)

    # ---- HIP ----------------------------------------------------------------------------------------------------------
    dev_batches = [helpers.to_device(b, DEV) for b in batches]

    def run(dropout_on):
        import itertools
        import vilbert.autograd_ops as AO
        orig = V._drop_p
        if not dropout_on:
            V._drop_p = lambda m: 0.0
        # the dropout masks are a function of (torch's seed, a per-process call counter): restart both, so that the statistical
        # bounds below see the same masks whatever ran before this test (round 6: the bounds tripped once inside the full suite
        # after new tests had moved the counter, and passed alone)
        torch.manual_seed(20260930)
        AO._seed_counter = itertools.count(1)
        try:
            net = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
            net.load_state_dict(sd)
            net = net.to(DEV).train()
            opt = AdamW(net.parameters(), lr=LR, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0)
            curve = []
            for step in range(STEPS):
                opt.zero_grad(set_to_none=True)
                loss = sum(l.mean() for l in net(*dev_batches[step % NB]))
                loss.backward()
                opt.step()
                curve.append(loss.item())
            with torch.no_grad():   # still in train mode, dropout probabilities as configured for this run
                repeat = [sum(l.mean() for l in net(*dev_batches[0])).item() for _ in range(2)]
        finally:
            V._drop_p = orig
        net.eval()
        with torch.no_grad():
            clean = sum(sum(l.mean() for l in net(*b)).item() for b in dev_batches) / NB
            evals = [sum(l.mean() for l in net(*dev_batches[0])).item() for _ in range(2)]
        return curve, clean, repeat, evals

    hip_off, clean_off, repeat_off, _ = run(False)
    hip_on, clean_on, repeat_on, eval_on = run(True)

    first, last = _window(oracle, 0, NB), _window(oracle, STEPS - 2 * NB, STEPS)
    assert last < 0.7 * first, "the oracle itself must learn on this problem (%.3f -> %.3f)" % (first, last)
    # no dropout: the same trajectory. Identical for the first steps (fp32 rounding only), then the two runs drift apart
    # slowly (different summation orders feed back through 200 Adam steps): window means within 2 % + 0.01.
    for i in range(3):
        assert abs(hip_off[i] - oracle[i]) <= 2e-4 * abs(oracle[i]), (i, hip_off[i], oracle[i])
    for lo in range(0, STEPS, 2 * NB):
        a, b = _window(hip_off, lo, lo + 2 * NB), _window(oracle, lo, lo + 2 * NB)
        assert abs(a - b) <= 0.02 * b + 0.01, "steps %d-%d: HIP %.4f vs oracle %.4f" % (lo, lo + 2 * NB, a, b)
    # dropout on: learns (same start, clear decrease) ...
    assert abs(hip_on[0] - oracle[0]) <= 0.15 * oracle[0]
    on_last = _window(hip_on, STEPS - 2 * NB, STEPS)
    assert on_last < 0.8 * _window(hip_on, 0, NB), (hip_on[:NB], on_last)
    # ... noisier and higher than the no-dropout curve at the end, but not by an order of magnitude (a missing 1 / (1 - p)
    # rescale or a wrong keep rate shows up here) ...
    assert last * 0.95 <= on_last <= 0.5 * first, (last, on_last, first)
    # ... and in eval mode (no masks) its weights fit the training batches about as well as the no-dropout run's
    assert clean_on <= 1.5 * clean_off + 0.5, (clean_on, clean_off)
    assert clean_on < 0.8 * first
    # fresh masks every step, none in eval mode - asked directly: the same batch twice WITHOUT an optimizer step in between
    # gives two different training-mode losses (a frozen mask would repeat the value) and identical eval-mode losses
    assert abs(repeat_on[0] - repeat_on[1]) > 1e-3 * abs(repeat_on[0]), repeat_on
    assert abs(repeat_off[0] - repeat_off[1]) <= 1e-6 * abs(repeat_off[0]), repeat_off
    assert abs(eval_on[0] - eval_on[1]) <= 1e-6 * abs(eval_on[0]), eval_on
