"""Seeded synthetic weights and inputs (TEST INFRASTRUCTURE, shared with bench.py).

The parameter table below restates the reference's ``state_dict`` ABI (names, shapes and
registration order of ``BertForMultiModalPreTraining`` / ``VILBertForVLTasks``,
reference vilbert/vilbert.py:320-367,396-533,536-694,697-900,1110-1258,1409-1432,1600-1722).
tests/test_oracle_vs_reference.py checks it key-for-key against the real reference.

Weights are generated per tensor from ``crc32(name) ^ seed`` so that any subset of the table
(e.g. ``bert.*`` shared by both wrappers) is reproducible on any machine with the same torch.
Distributions follow SURVEY.md section 8(d): matrices ~ N(0, 0.02); biases ~ N(0, 0.02) and
LayerNorm gamma ~ 1 + N(0, 0.1), beta ~ N(0, 0.1) (the constructor defaults of 0 / 1 would
hide bias and affine bugs).
"""
import json
import os
import zlib

import torch

CONFIG_DIR = os.path.join(
    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vilbert-multi-task_amd", "config"
)

_DEFAULTS = dict(
    hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
    hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
    max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02, v_feature_size=2048,
    v_target_size=1601, v_hidden_size=768, v_num_hidden_layers=3, v_num_attention_heads=12,
    v_intermediate_size=3072, bi_hidden_size=1024, bi_num_attention_heads=16,
    v_attention_probs_dropout_prob=0.1, v_hidden_act="gelu", v_hidden_dropout_prob=0.1,
    v_initializer_range=0.2, v_biattention_id=[0, 1], t_biattention_id=[10, 11], visual_target=0,
    fast_mode=False, fixed_v_layer=0, fixed_t_layer=0, in_batch_pairs=False, fusion_method="mul",
    dynamic_attention=False, with_coattention=True, objective=0, num_negative=128, model="bert",
    task_specific_tokens=False, visualization=False,
)


def load_config(name_or_path):
    """Config JSON -> plain dict with the reference's constructor defaults filled in
    (reference vilbert/vilbert.py:270-282: defaults first, JSON keys override)."""
    path = name_or_path if os.path.isfile(name_or_path) else os.path.join(CONFIG_DIR, name_or_path)
    with open(path, "r", encoding="utf-8") as f:
        cfg = dict(_DEFAULTS, vocab_size=-1)
        cfg.update(json.load(f))
    return cfg


def tiny_config(**over):
    """A small two-stream config (all code paths, seconds on CPU) for unit tests."""
    cfg = dict(_DEFAULTS)
    cfg.update(
        vocab_size=97, hidden_size=64, num_hidden_layers=3, num_attention_heads=2,
        intermediate_size=128, max_position_embeddings=40, v_feature_size=48, v_target_size=11,
        v_hidden_size=96, v_num_hidden_layers=2, v_num_attention_heads=3, v_intermediate_size=80,
        bi_hidden_size=64, bi_num_attention_heads=2, v_biattention_id=[0, 1],
        t_biattention_id=[1, 2],
    )
    cfg.update(over)
    return cfg


def _lin(name, out_f, in_f):
    return [(name + ".weight", (out_f, in_f), "w"), (name + ".bias", (out_f,), "b")]


def _ln(name, n):
    return [(name + ".weight", (n,), "g"), (name + ".bias", (n,), "beta")]


def param_table(cfg, kind):
    """[(name, shape, init)] in ``state_dict()`` order. kind: 'pretraining' | 'vltasks'."""
    H, I, Hv, Iv, Hb = (cfg["hidden_size"], cfg["intermediate_size"], cfg["v_hidden_size"],
                        cfg["v_intermediate_size"], cfg["bi_hidden_size"])
    V = cfg["vocab_size"]
    t = []
    e = "bert.embeddings."
    t += [(e + "word_embeddings.weight", (V, H), "w"),
          (e + "position_embeddings.weight", (cfg["max_position_embeddings"], H), "w"),
          (e + "token_type_embeddings.weight", (cfg["type_vocab_size"], H), "w")]
    t += _ln(e + "LayerNorm", H)
    if cfg["task_specific_tokens"]:
        t += [(e + "task_embeddings.weight", (20, H), "w")]
    ve = "bert.v_embeddings."
    t += _lin(ve + "image_embeddings", Hv, cfg["v_feature_size"])
    t += _lin(ve + "image_location_embeddings", Hv, 5)
    t += _ln(ve + "LayerNorm", Hv)

    def stream_layer(p, h, inter, hid_txt=None):
        r = []
        for n in ("query", "key", "value"):
            r += _lin(p + "attention.self." + n, h, h)
        if hid_txt is not None and cfg["dynamic_attention"]:
            r += _lin(p + "attention.self.dyLinear_q", h, hid_txt)
            r += _lin(p + "attention.self.dyLinear_k", h, hid_txt)
        r += _lin(p + "attention.output.dense", h, h) + _ln(p + "attention.output.LayerNorm", h)
        r += _lin(p + "intermediate.dense", inter, h)
        r += _lin(p + "output.dense", h, inter) + _ln(p + "output.LayerNorm", h)
        return r

    for i in range(cfg["num_hidden_layers"]):
        t += stream_layer("bert.encoder.layer.%d." % i, H, I)
    for i in range(cfg["v_num_hidden_layers"]):
        t += stream_layer("bert.encoder.v_layer.%d." % i, Hv, Iv, hid_txt=H)
    for i in range(len(cfg["v_biattention_id"])):
        p = "bert.encoder.c_layer.%d." % i
        for n in ("query1", "key1", "value1"):
            t += _lin(p + "biattention." + n, Hb, Hv)
        for n in ("query2", "key2", "value2"):
            t += _lin(p + "biattention." + n, Hb, H)
        t += _lin(p + "biOutput.dense1", Hv, Hb) + _ln(p + "biOutput.LayerNorm1", Hv)
        t += _lin(p + "biOutput.q_dense1", Hv, Hb)
        t += _lin(p + "biOutput.dense2", H, Hb) + _ln(p + "biOutput.LayerNorm2", H)
        t += _lin(p + "biOutput.q_dense2", H, Hb)
        t += _lin(p + "v_intermediate.dense", Iv, Hv)
        t += _lin(p + "v_output.dense", Hv, Iv) + _ln(p + "v_output.LayerNorm", Hv)
        t += _lin(p + "t_intermediate.dense", I, H)
        t += _lin(p + "t_output.dense", H, I) + _ln(p + "t_output.LayerNorm", H)
    t += _lin("bert.t_pooler.dense", Hb, H) + _lin("bert.v_pooler.dense", Hb, Hv)

    t += [("cls.predictions.bias", (V,), "b")]
    t += _lin("cls.predictions.transform.dense", H, H) + _ln("cls.predictions.transform.LayerNorm", H)
    t += [("cls.predictions.decoder.weight", (V, H), "tied:bert.embeddings.word_embeddings.weight")]
    t += _lin("cls.bi_seq_relationship", 2, Hb)
    t += _lin("cls.imagePredictions.transform.dense", Hv, Hv)
    t += _ln("cls.imagePredictions.transform.LayerNorm", Hv)
    t += _lin("cls.imagePredictions.decoder", cfg["v_target_size"], Hv)
    if kind == "vltasks":
        def classifier(p, i, h, o):
            return _lin(p + ".logit_fc.0", h, i) + _ln(p + ".logit_fc.2", h) + _lin(p + ".logit_fc.3", o, h)
        t += classifier("vil_prediction", Hb, 2 * Hb, 3129)
        t += classifier("vil_prediction_gqa", Hb, 2 * Hb, 1533)
        t += classifier("vil_binary_prediction", 2 * Hb, 2 * Hb, 2)
        t += _lin("vil_logit", 1, Hb) + _lin("vil_tri_prediction", 3, Hb)
        t += _lin("vision_logit", 1, Hv) + _lin("linguisic_logit", 1, H)
    elif kind != "pretraining":
        raise ValueError(kind)
    return t


def make_state_dict(cfg, kind, seed=1234, dtype=torch.float32):
    sd = {}
    for name, shape, init in param_table(cfg, kind):
        if init.startswith("tied:"):
            sd[name] = sd[init[5:]]
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
        x = torch.randn(shape, generator=g, dtype=torch.float32)
        if init in ("w", "b"):
            x.mul_(0.02)
        elif init == "g":
            x.mul_(0.1).add_(1.0)
        elif init == "beta":
            x.mul_(0.1)
        sd[name] = x.to(dtype)
    return sd


def make_inputs(cfg, batch, n_tok, n_reg, seed=7, ragged=True, with_labels=False, task_id=None):
    """Synthetic batch following SURVEY.md section 8(d). Returns a dict of CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    V, F = cfg["vocab_size"], cfg["v_feature_size"]
    ids = torch.randint(0, V, (batch, n_tok), generator=g)
    ids[:, 0] = min(101, V - 1)
    seg = torch.zeros(batch, n_tok, dtype=torch.long)
    if ragged:
        lo_t, lo_r = min(8, n_tok), min(10, n_reg)
        tl = torch.randint(lo_t, n_tok + 1, (batch,), generator=g)
        rl = torch.randint(lo_r, n_reg + 1, (batch,), generator=g)
    else:
        tl = torch.full((batch,), n_tok)
        rl = torch.full((batch,), n_reg)
    mask = (torch.arange(n_tok)[None, :] < tl[:, None]).long()
    imask = (torch.arange(n_reg)[None, :] < rl[:, None]).long()
    feat = torch.rand(batch, n_reg, F, generator=g) * 2.0
    feat = feat * imask[:, :, None].float()
    loc = torch.rand(batch, n_reg, 5, generator=g)
    loc[:, 0] = torch.tensor([0.0, 0.0, 1.0, 1.0, 1.0])
    out = dict(input_ids=ids, image_feat=feat, image_loc=loc, token_type_ids=seg,
               attention_mask=mask, image_attention_mask=imask,
               co_attention_mask=torch.zeros(batch, n_reg, n_tok))
    if task_id is not None:
        out["task_ids"] = torch.full((batch, 1), task_id, dtype=torch.long)
    if with_labels:
        # train_concap loader conventions (reference vilbert/datasets/concept_cap_dataset.py:
        # 244-282,608-670): ~15 % of tokens / regions carry a label, -1 elsewhere; image_target
        # rows are probability vectors; region 0 is the global feature and has no target.
        lm = torch.where(torch.rand(batch, n_tok, generator=g) < 0.15, ids, torch.full_like(ids, -1))
        lm = torch.where(mask.bool(), lm, torch.full_like(lm, -1))
        lm[:, 1] = ids[:, 1]  # at least one labelled token per row
        il = torch.where(torch.rand(batch, n_reg - 1, generator=g) < 0.15, 1, -1)
        il[:, 0] = 1  # at least one labelled region (the reference divides by the count)
        tgt = torch.softmax(torch.randn(batch, n_reg - 1, cfg["v_target_size"], generator=g), -1)
        out.update(masked_lm_labels=lm, image_label=il, image_target=tgt,
                   next_sentence_label=torch.randint(0, 2, (batch,), generator=g))
    return out
