"""Parity at BASELINE.json's full single-GPU size (bert_base_6layer_6conect, batch 256, 36 tokens x 36 regions)
- every row of every output against the CPU oracle (test_full_size_forward_all_rows_match_oracle, ~10 s of host
time), and through size-independent properties:
  * samples 0-1 of the batch ARE the golden case base_6l6c_b2 (outputs of the real reference): the encoder is
    per-sample independent, so their rows of the batch-256 outputs must match the golden vectors at the 1e-4 bar
    (different GEMM tiling: M = 9216 rows, hybrid tail, all 256 CUs);
  * batch permutation: rolling the batch rolls every output;
  * additivity of the backward: with a fixed cotangent the parameter gradients of the batch equal the sum of
    the gradients of its two halves (exercises the split-K wgrad at the full contraction length)."""
import pytest
import torch

import helpers
from helpers import cases
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, T, R = 256, 36, 36


def _batch():
    cfg, sd, x2 = cases.case_inputs("base_6l6c_b2")
    x = synth.make_inputs(cfg, B, T, R, seed=99, ragged=True)
    for k in x2:
        x[k][:2] = x2[k]
    return cfg, sd, x


def _args(x):
    return (x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
            x["image_attention_mask"], x["co_attention_mask"])


def test_full_size_forward_rows_match_golden_and_batch_permutation():
    from vilbert.vilbert import BertConfig, VILBertForVLTasks
    cfg, sd, x = _batch()
    model = VILBertForVLTasks(BertConfig.from_dict(cfg), num_labels=1)
    model.load_state_dict(sd, strict=True)
    model = model.eval().to(DEV)
    with torch.no_grad():
        out = model(*helpers.to_device(_args(x), DEV))
        gold = helpers.load_golden("base_6l6c_b2")
        for i, n in enumerate(cases.VL_NAMES):
            rows = 1 if n == "vil_binary_prediction" else 2        # pairs (2i, 2i + 1) share one row
            helpers.assert_close(cases.sample("base_6l6c_b2", n, out[i][:rows]), gold[n], "B=256 rows / " + n)
        shift = 128
        rolled = tuple(a.roll(shift, 0) for a in _args(x))
        out2 = model(*helpers.to_device(rolled, DEV))
        for i, n in enumerate(cases.VL_NAMES):
            s = shift // 2 if n == "vil_binary_prediction" else shift
            helpers.assert_close(out2[i], out[i].roll(s, 0), "rolled / " + n, atol=2e-5, rtol=2e-5)


@pytest.mark.slow
def test_full_size_forward_all_rows_match_oracle():
    """BASELINE.json configs[1] in full: bert_base_6layer_6conect forward, batch 256, 36 x 36, ragged masks -
    ALL 256 rows of all nine VILBertForVLTasks outputs against oracle/vilbert_oracle.py at the 1e-4 bar."""
    from vilbert.vilbert import BertConfig, VILBertForVLTasks
    from oracle import vilbert_oracle as vo
    cfg, sd, x = _batch()
    model = VILBertForVLTasks(BertConfig.from_dict(cfg), num_labels=1)
    model.load_state_dict(sd, strict=True)
    model = model.eval().to(DEV)
    with torch.no_grad():
        out = model(*helpers.to_device(_args(x), DEV))
        for lo in range(0, B, 64):           # the oracle in four chunks (its [64, 36, 30522] logits are 0.28 GB each)
            want = vo.vltasks_forward(sd, cfg, *(a[lo:lo + 64] for a in _args(x)))
            for i, n in enumerate(cases.VL_NAMES):
                rows = slice(lo // 2, (lo + 64) // 2) if n == "vil_binary_prediction" else slice(lo, lo + 64)
                helpers.assert_close(out[i][rows], want[i], "B=256 rows %d.. / %s" % (lo, n))


def test_full_size_backward_is_additive_over_the_batch():
    from vilbert.vilbert import BertConfig, VILBertForVLTasks
    cfg, sd, x = _batch()
    model = VILBertForVLTasks(BertConfig.from_dict(cfg), num_labels=1)
    model.load_state_dict(sd, strict=True)
    model = model.eval().to(DEV)          # eval: dropout off, autograd still records
    bert = model.bert
    args = helpers.to_device(_args(x), DEV)
    g = torch.Generator().manual_seed(5)
    cot = [torch.randn(B, T, cfg["hidden_size"], generator=g).to(DEV) * 1e-2,
           torch.randn(B, R, cfg["v_hidden_size"], generator=g).to(DEV) * 1e-2,
           torch.randn(B, cfg["bi_hidden_size"], generator=g).to(DEV) * 1e-2,
           torch.randn(B, cfg["bi_hidden_size"], generator=g).to(DEV) * 1e-2]

    def grads(lo, hi):
        bert.zero_grad(set_to_none=True)
        sl = tuple(a[lo:hi] for a in args)
        seq_t, seq_v, pool_t, pool_v = bert(sl[0], sl[1], sl[2], sl[3], sl[4], sl[5], sl[6])[:4]
        loss = sum((o * c[lo:hi]).sum() for o, c in zip((seq_t, seq_v, pool_t, pool_v), cot))
        loss.backward()
        return {n: p.grad.detach().clone() for n, p in bert.named_parameters() if p.grad is not None}

    whole = grads(0, B)
    first, second = grads(0, B // 2), grads(B // 2, B)
    assert set(whole) == set(first) == set(second) and len(whole) > 300
    gmax = max(float(v.abs().max()) for v in whole.values())
    for n, w in whole.items():
        parts = first[n] + second[n]
        err = float((w - parts).abs().max())
        bound = 2e-4 * float(w.abs().max()) + 2e-7 * gmax
        assert err <= bound, "%s: |whole - (half + half)| = %.3e > %.3e" % (n, err, bound)
