"""Host-side boundary (CPU): BertConfig semantics, state_dict ABI, weight tying, from_pretrained."""
import json
import os

import pytest
import torch

from oracle import ref_loader, synth
from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining, VILBertForVLTasks

CONFIG_DIR = synth.CONFIG_DIR


def _build(kind, cfg):
    c = BertConfig.from_dict(cfg)
    return VILBertForVLTasks(c, num_labels=1) if kind == "vltasks" else BertForMultiModalPreTraining(c)


@pytest.mark.parametrize("kind", ["pretraining", "vltasks"])
@pytest.mark.parametrize("over", [{}, {"task_specific_tokens": True}, {"dynamic_attention": True},
                                  {"model": "roberta"}])
def test_state_dict_names_shapes_order(kind, over):
    cfg = synth.tiny_config(**over)
    got = [(k, tuple(v.shape)) for k, v in _build(kind, cfg).state_dict().items()]
    assert got == [(n, tuple(s)) for n, s, _ in synth.param_table(cfg, kind)]


def test_named_parameters_match_reference_names():
    if not ref_loader.available():
        pytest.skip("reference tree not present")
    ref = ref_loader.load()
    cfg = synth.tiny_config(task_specific_tokens=True)
    ours = VILBertForVLTasks(BertConfig.from_dict(cfg), 1)
    theirs = ref.VILBertForVLTasks(ref.BertConfig.from_dict(cfg), 1)
    assert [(n, tuple(p.shape)) for n, p in ours.named_parameters()] == \
        [(n, tuple(p.shape)) for n, p in theirs.named_parameters()]
    assert [n for n, _ in ours.named_modules()] == [n for n, _ in theirs.named_modules()]


def test_config_json_semantics():
    c = BertConfig.from_json_file(os.path.join(CONFIG_DIR, "bert_base_6layer_6conect.json"))
    assert c.v_biattention_id == [0, 1, 2, 3, 4, 5] and c.t_biattention_id == [6, 7, 8, 9, 10, 11]
    assert c.bi_hidden_size == 1024 and c.v_hidden_size == 1024 and c.hidden_size == 768
    # keys absent from the JSON keep constructor defaults; unread keys are tolerated and kept
    assert c.fusion_method == "mul" and c.with_coattention is True and c.task_specific_tokens is False
    assert hasattr(c, "bi_intermediate_size")
    rt = BertConfig.from_dict(json.loads(c.to_json_string()))
    assert rt.to_dict() == c.to_dict()
    assert repr(c) == c.to_json_string()
    d = BertConfig(30522)
    assert d.vocab_size == 30522 and d.hidden_size == 768 and d.v_feature_size == 2048
    with pytest.raises(ValueError):
        BertConfig(3.5)
    with pytest.raises(AssertionError):
        BertConfig(10, v_biattention_id=[0, 5], v_num_hidden_layers=3)


def test_config_matches_reference_defaults():
    if not ref_loader.available():
        pytest.skip("reference tree not present")
    ref = ref_loader.load()
    assert BertConfig(123).to_dict() == ref.BertConfig(123).to_dict()
    for name in sorted(os.listdir(CONFIG_DIR)):
        if name.endswith(".json") and "6conect" in name:
            ours = BertConfig.from_json_file(os.path.join(CONFIG_DIR, name)).to_dict()
            theirs = ref.BertConfig.from_json_file(ref_loader.config_path(name)).to_dict()
            assert ours == theirs, name


def test_head_divisibility_errors():
    with pytest.raises(ValueError):
        _build("vltasks", synth.tiny_config(hidden_size=65))
    with pytest.raises(ValueError):
        _build("vltasks", synth.tiny_config(bi_hidden_size=65))


def test_unsupported_head_size_is_reported_at_construction():
    """Head sizes outside the compiled set {32, 64, 128} are refused when the model is built, not inside a forward."""
    with pytest.raises(NotImplementedError, match="head size"):
        _build("vltasks", synth.tiny_config(num_attention_heads=4))          # 64 / 4 = 16
    with pytest.raises(NotImplementedError, match="head size"):
        _build("vltasks", synth.tiny_config(v_hidden_size=96, v_num_attention_heads=2))   # 48


def test_weight_tying_and_init():
    m = _build("pretraining", synth.tiny_config())
    assert m.cls.predictions.decoder.weight is m.bert.embeddings.word_embeddings.weight
    # reference init (vilbert.py:1274-1285): zero biases, unit LayerNorm
    assert float(m.bert.encoder.layer[0].attention.self.query.bias.abs().sum()) == 0.0
    assert float((m.bert.embeddings.LayerNorm.weight - 1).abs().sum()) == 0.0
    sd = m.state_dict()
    assert sd["cls.predictions.decoder.weight"].data_ptr() == sd["bert.embeddings.word_embeddings.weight"].data_ptr()


def test_from_pretrained(tmp_path):
    cfg = synth.tiny_config()
    config = BertConfig.from_dict(cfg)
    assert VILBertForVLTasks.from_pretrained(str(tmp_path / "missing.bin"), config=config, num_labels=1) is None
    sd = synth.make_state_dict(cfg, "pretraining")
    # old-style LayerNorm names are renamed on load
    old = {k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta"): v
           for k, v in sd.items()}
    path = tmp_path / "pytorch_model.bin"
    torch.save(old, str(path))
    m, info = VILBertForVLTasks.from_pretrained(str(path), config=config, num_labels=1, default_gpu=False,
                                                output_loading_info=True)
    assert not m.training
    assert all(k.startswith(("vil_", "vision_logit", "linguisic_logit")) for k in info["missing_keys"])
    assert torch.equal(m.bert.encoder.layer[1].output.LayerNorm.weight, sd["bert.encoder.layer.1.output.LayerNorm.weight"])
    assert m.cls.predictions.decoder.weight is m.bert.embeddings.word_embeddings.weight
    # a bare BertModel checkpoint (no "bert." prefix) loads into the wrapper's base model
    bare = {k[len("bert."):]: v for k, v in sd.items() if k.startswith("bert.")}
    m2 = BertForMultiModalPreTraining.from_pretrained(str(tmp_path), config=config, state_dict=bare)
    assert torch.equal(m2.bert.t_pooler.dense.weight, sd["bert.t_pooler.dense.weight"])


def test_constructor_and_forward_signatures_equal_the_reference():
    """Drop-in boundary (round-2 verdict, harness item 13): every public class of the hot path has the reference's
    ``__init__`` and ``forward`` signature - parameter names, order, kinds and defaults - compared with
    inspect.signature against the REAL reference classes (build container only; the GPU box has no reference tree)."""
    import inspect
    if not ref_loader.available():
        pytest.skip("reference tree not present")
    ref = ref_loader.load()
    import vilbert.vilbert as ours
    names = ["BertConfig", "BertEmbeddings", "BertImageEmbeddings", "BertSelfAttention", "BertSelfOutput", "BertAttention",
             "BertIntermediate", "BertOutput", "BertLayer", "BertImageSelfAttention", "BertImageSelfOutput",
             "BertImageAttention", "BertImageIntermediate", "BertImageOutput", "BertImageLayer", "BertBiAttention",
             "BertBiOutput", "BertConnectionLayer", "BertEncoder", "BertTextPooler", "BertImagePooler", "BertModel",
             "BertPreTrainingHeads", "BertForMultiModalPreTraining", "VILBertForVLTasks", "SimpleClassifier"]
    checked = 0
    for name in names:
        theirs, mine = getattr(ref, name, None), getattr(ours, name, None)
        if theirs is None:
            continue
        assert mine is not None, "missing class " + name
        for meth in ("__init__", "forward"):
            if not hasattr(theirs, meth) or getattr(theirs, meth) is getattr(object, meth, None):
                continue
            st, sm = inspect.signature(getattr(theirs, meth)), inspect.signature(getattr(mine, meth))
            got = [(p.name, p.kind, p.default) for p in sm.parameters.values()]
            want = [(p.name, p.kind, p.default) for p in st.parameters.values()]
            assert got == want, "%s.%s: %s != reference %s" % (name, meth, sm, st)
            checked += 1
    assert checked >= 40
    for fn in ("from_dict", "from_json_file", "to_dict", "to_json_string"):
        assert str(inspect.signature(getattr(ours.BertConfig, fn))) == str(inspect.signature(getattr(ref.BertConfig, fn)))


def test_half_switches_this_model_to_the_bf16_mode_and_keeps_fp32_master_weights():
    """The reference's scripts call `model.half()` for reduced-precision training (train_concap.py:504-505) next to an
    optimizer that keeps fp32 master weights; here that is the bf16 stream, PER MODEL (round 6): the model's forward runs in
    the bf16 mode and restores the process-wide mode, a second model is unaffected, `float()` undoes it; the parameters stay
    fp32 (no kernel of the package takes fp16 parameters), the call returns the model like nn.Module.half()."""
    from vilbert import _native
    m, other = _build("pretraining", synth.tiny_config()), _build("pretraining", synth.tiny_config())
    prev = _native.set_gemm_mode("f32")
    try:
        seen = []
        for net in (m, other):
            net.forward = lambda *a, **k: seen.append(_native.bf16_stream())      # (no device here: only the mode is observed)
        assert m.half() is m
        assert not _native.bf16_stream(), "half() must not flip the process-wide mode"
        m()
        other()
        assert seen == [True, False] and not _native.bf16_stream()
        assert all(p.dtype == torch.float32 for p in m.parameters())
        _native.set_gemm_mode("mxfp8")           # another mode active in the process: restored after the half() model's forward
        m()
        assert _native.set_gemm_mode("f32") == "mxfp8"
        assert m.float() is m
        m()
        assert seen[-1] is False
    finally:
        _native.set_gemm_mode(prev)
