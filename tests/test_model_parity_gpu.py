"""Model-level parity on the GPU, through the reference-shaped Python API:
  * against the committed golden vectors (outputs of the REAL reference, tests/golden/),
  * against the CPU oracle on further seeded inputs (ragged masks, all-ones masks, other seeds).
Bar: |got - want| <= 1e-4 + 1e-4 |want| on every output (BASELINE.json north_star: fp32 1e-4)."""
import pytest
import torch

import helpers
from helpers import cases
from oracle import synth, vilbert_oracle as vo

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(kind, cfg, sd):
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining, VILBertForVLTasks
    c = BertConfig.from_dict(cfg)
    m = VILBertForVLTasks(c, num_labels=1) if kind == "vltasks" else BertForMultiModalPreTraining(c)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return m.eval().to(DEV)


@pytest.mark.parametrize("case", list(cases.CASES))
def test_hip_model_matches_reference_golden(case):
    c = cases.CASES[case]
    cfg, sd, x = cases.case_inputs(case)
    model = _model(c["kind"], cfg, sd)
    with torch.no_grad():
        out = model(*helpers.to_device(cases.forward_args(case, x), DEV))
    gold = helpers.load_golden(case)
    for i, n in enumerate(cases.output_names(case)):
        helpers.assert_close(cases.sample(case, n, out[i]), gold[n], "%s/%s" % (case, n))


@pytest.mark.parametrize("cfgname,batch,n_tok,n_reg,seed,ragged", [
    ("tiny", 5, 12, 10, 11, True),
    ("tiny", 2, 1, 1, 12, False),              # single token / single region
    ("tiny", 3, 40, 37, 13, True),             # max_position_embeddings of the tiny config
    ("bert_base_6layer_6conect.json", 8, 36, 36, 14, True),
    ("bert_base_6layer_6conect.json", 4, 36, 37, 15, False),   # train_concap shape (36 + 1 global region)
    ("bert_base_6layer_6conect.json", 2, 23, 101, 16, True),   # VQA task shape from vilbert_tasks.yml
])
def test_hip_model_matches_oracle(cfgname, batch, n_tok, n_reg, seed, ragged):
    cfg = synth.tiny_config() if cfgname == "tiny" else synth.load_config(cfgname)
    sd = synth.make_state_dict(cfg, "vltasks", seed=seed)
    x = synth.make_inputs(cfg, batch, n_tok, n_reg, seed=seed, ragged=ragged)
    args = (x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
            x["image_attention_mask"], x["co_attention_mask"])
    model = _model("vltasks", cfg, sd)
    with torch.no_grad():
        got = model(*helpers.to_device(args, DEV))
        want = vo.vltasks_forward(sd, cfg, *args)
    for n, g, w in zip(cases.VL_NAMES, got, want):
        helpers.assert_close(g, w, n)


def test_bert_model_outputs_and_optional_arguments():
    cfg = synth.tiny_config()
    sd = synth.make_state_dict(cfg, "vltasks")
    x = synth.make_inputs(cfg, 3, 8, 6, ragged=False)
    model = _model("vltasks", cfg, sd).bert
    dev = lambda t: t.to(DEV)
    with torch.no_grad():
        # masks / token types default to ones / zeros like the reference (vilbert.py:1322-1329)
        seq_t, seq_v, pool_t, pool_v, attn = model(dev(x["input_ids"]), dev(x["image_feat"]), dev(x["image_loc"]))
        w_t, w_v, wp_t, wp_v, _ = vo.bert_model(sd, cfg, x["input_ids"], x["image_feat"], x["image_loc"])
        all_t, all_v, _, _, _ = model(dev(x["input_ids"]), dev(x["image_feat"]), dev(x["image_loc"]),
                                      output_all_encoded_layers=True)
        wl_t, wl_v, _, _, _ = vo.bert_model(sd, cfg, x["input_ids"], x["image_feat"], x["image_loc"],
                                            all_layers=True)
    for n, g, w in (("seq_t", seq_t, w_t), ("seq_v", seq_v, w_v), ("pooled_t", pool_t, wp_t), ("pooled_v", pool_v, wp_v)):
        helpers.assert_close(g, w, n)
    assert attn == ([], [], [])
    assert len(all_t) == len(wl_t) == len(cfg["v_biattention_id"])
    for g, w in zip(all_t + all_v, wl_t + wl_v):
        helpers.assert_close(g, w, "per-connection outputs")


def test_visualization_returns_attention_probabilities():
    cfg = synth.tiny_config(visualization=True)
    sd = synth.make_state_dict(cfg, "vltasks")
    x = synth.make_inputs(cfg, 2, 7, 5)
    args = (x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
            x["image_attention_mask"], x["co_attention_mask"])
    model = _model("vltasks", cfg, sd)
    with torch.no_grad():
        out = model(*helpers.to_device(args, DEV), None, False, True)
        _, _, _, _, (wt, wv, wc) = vo.bert_model(sd, cfg, *args)
    at, av, ac = out[9]
    assert len(at) == cfg["num_hidden_layers"] and len(av) == cfg["v_num_hidden_layers"] and len(ac) == 2
    for g, w in zip(at + av, wt + wv):
        helpers.assert_close(g["attn"], w, "self-attention probs", atol=1e-5)
        assert g["queries"].shape == g["keys"].shape == w.shape[:3] + (g["queries"].shape[-1],)
    for g, (p1, p2) in zip(ac, wc):
        helpers.assert_close(g["attn1"], p1, "co-attention probs 1", atol=1e-5)
        helpers.assert_close(g["attn2"], p2, "co-attention probs 2", atol=1e-5)


def test_pretraining_losses_match_oracle():
    cfg = synth.tiny_config()
    sd = synth.make_state_dict(cfg, "pretraining")
    x = synth.make_inputs(cfg, 4, 9, 8, with_labels=True)
    args = (x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
            x["image_attention_mask"], x["masked_lm_labels"], x["image_label"], x["image_target"],
            x["next_sentence_label"])
    model = _model("pretraining", cfg, sd)
    with torch.no_grad():
        got = model(*helpers.to_device(args, DEV))
        want = vo.pretraining_forward(sd, cfg, *args)
    for n, g, w in zip(cases.LOSS_NAMES, got, want):
        assert g.shape == (1,)
        helpers.assert_close(g, w, n)


def test_smoke_entry_point():
    import __graft_entry__
    __graft_entry__.smoke()
