"""Pins oracle/ (restatement + parameter table) against the REAL reference model.

Runs only where /root/reference exists (the build container); on the GPU box the committed
fixtures in tests/golden/ (made from the same reference by tests/golden/make_golden.py) pin it.
"""
import pytest
import torch

from oracle import ref_loader, synth, vilbert_oracle as vo

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")


def _ref_model(cfg, kind, sd):
    ref = ref_loader.load()
    rc = ref.BertConfig.from_dict(cfg)
    m = ref.BertForMultiModalPreTraining(rc) if kind == "pretraining" else ref.VILBertForVLTasks(rc, num_labels=1)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.eval()


@pytest.mark.parametrize("kind", ["pretraining", "vltasks"])
@pytest.mark.parametrize("cfgname", ["tiny", "tiny_task", "bert_base_2layer_2conect.json"])
def test_param_table_matches_reference_state_dict(kind, cfgname):
    ref = ref_loader.load()
    cfg = (synth.tiny_config() if cfgname == "tiny" else synth.tiny_config(task_specific_tokens=True)
           if cfgname == "tiny_task" else synth.load_config(cfgname))
    if cfgname.endswith(".json"):
        cfg = dict(cfg, num_hidden_layers=2, t_biattention_id=[0, 1])  # keep the ctor cheap
    rc = ref.BertConfig.from_dict(cfg)
    m = ref.BertForMultiModalPreTraining(rc) if kind == "pretraining" else ref.VILBertForVLTasks(rc, 1)
    want = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    got = [(n, tuple(s)) for n, s, _ in synth.param_table(cfg, kind)]
    assert got == want


def _close(a, b, tol):
    assert a.shape == b.shape
    err = (a - b).abs().max().item()
    assert err <= tol, err


def test_vltasks_forward_matches_reference():
    cfg = synth.tiny_config()
    sd = synth.make_state_dict(cfg, "vltasks")
    x = synth.make_inputs(cfg, 4, 9, 7)
    args = (x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
            x["image_attention_mask"], x["co_attention_mask"])
    with torch.no_grad():
        want = _ref_model(cfg, "vltasks", sd)(*args)[:9]
        got = vo.vltasks_forward(sd, cfg, *args)
    for g, w in zip(got, want):
        _close(g, w, 2e-6 * max(1.0, w.abs().max().item()))


def test_task_tokens_and_odd_batch():
    cfg = synth.tiny_config(task_specific_tokens=True)
    sd = synth.make_state_dict(cfg, "vltasks")
    x = synth.make_inputs(cfg, 3, 6, 5, task_id=4)
    args = (x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
            x["image_attention_mask"], x["co_attention_mask"], x["task_ids"])
    with torch.no_grad():
        want = _ref_model(cfg, "vltasks", sd)(*args)[:9]
        got = vo.vltasks_forward(sd, cfg, *args)
    for g, w in zip(got, want):
        _close(g, w, 2e-6 * max(1.0, w.abs().max().item()))


@pytest.mark.parametrize("visual_target", [0, 1])
def test_pretraining_losses_and_grads_match_reference(visual_target):
    cfg = synth.tiny_config(visual_target=visual_target, hidden_dropout_prob=0.0,
                            attention_probs_dropout_prob=0.0, v_hidden_dropout_prob=0.0,
                            v_attention_probs_dropout_prob=0.0)
    sd = synth.make_state_dict(cfg, "pretraining")
    x = synth.make_inputs(cfg, 4, 9, 8, with_labels=True)
    if visual_target == 1:
        x["image_target"] = torch.randn(4, 7, cfg["v_target_size"])
    args = (x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
            x["image_attention_mask"], x["masked_lm_labels"], x["image_label"], x["image_target"],
            x["next_sentence_label"])
    m = _ref_model(cfg, "pretraining", sd)
    m.cls.dropout.p = 0.0
    m.train()
    ref_losses = m(*args)
    sum(l.sum() for l in ref_losses).backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not k.endswith("decoder.weight") or "image" in k}
    leaves["cls.predictions.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]
    got = vo.pretraining_forward(leaves, cfg, *args)
    sum(l.sum() for l in got).backward()
    for g, w in zip(got, ref_losses):
        _close(g.detach(), w.detach(), 1e-5)
    for name, p in m.named_parameters():
        if p.grad is None:
            assert leaves[name].grad is None or leaves[name].grad.abs().max() == 0, name
        else:
            _close(leaves[name].grad, p.grad, 1e-5 * max(1.0, p.grad.abs().max().item()))
