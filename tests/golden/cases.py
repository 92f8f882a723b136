"""Golden-vector cases shared by make_golden.py (which runs the REAL reference) and the tests.

Weights and inputs are not stored: both are regenerated from seeds by oracle/synth.py (per-tensor
seeded torch CPU generators), only the reference's outputs are committed (tests/golden/*.npz).
Large outputs are stored as strided samples (`stride` along the last dim) to keep fixtures small.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import synth  # noqa: E402

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))

VL_NAMES = ["vil_prediction", "vil_prediction_gqa", "vil_logit", "vil_binary_prediction", "vil_tri_prediction",
            "vision_prediction", "vision_logit", "linguisic_prediction", "linguisic_logit"]
PT_NAMES = ["prediction_scores_t", "prediction_scores_v", "seq_relationship_score"]
LOSS_NAMES = ["masked_lm_loss", "masked_img_loss", "next_sentence_loss"]

# name -> dict(kind, cfg (callable), batch, n_tok, n_reg, task_id, labels, stride{output: step})
CASES = {
    "tiny_vltasks": dict(kind="vltasks", cfg=lambda: synth.tiny_config(), batch=4, n_tok=9, n_reg=7),
    "tiny_task_tokens_odd_batch": dict(kind="vltasks", cfg=lambda: synth.tiny_config(task_specific_tokens=True),
                                       batch=3, n_tok=6, n_reg=5, task_id=4),
    "tiny_dynamic_attention_sum": dict(kind="vltasks",
                                       cfg=lambda: synth.tiny_config(dynamic_attention=True, fusion_method="sum"),
                                       batch=2, n_tok=5, n_reg=6),
    "tiny_pretraining_scores": dict(kind="pretraining", cfg=lambda: synth.tiny_config(), batch=4, n_tok=9, n_reg=8),
    "tiny_pretraining_losses": dict(kind="pretraining", cfg=lambda: synth.tiny_config(), batch=4, n_tok=9, n_reg=8,
                                    labels=True),
    # inference variants of the same path
    # eval_retrieval.py:220,263-311: fast_mode, ONE caption against N images (text batch 1 is expanded
    # inside the encoder, reference vilbert.py:1042-1053)
    "tiny_fast_mode_1xN": dict(kind="vltasks", cfg=lambda: synth.tiny_config(fast_mode=True), batch=5, n_tok=8,
                               n_reg=6, text_batch=1),
    # in_batch_pairs: every caption against every image, batch becomes B*B (:1008-1040)
    # (pre-training wrapper: VILBertForVLTasks itself cannot run this mode in the reference - its
    # vision_logit mask is still batch B, vilbert.py:1692-1694)
    "tiny_in_batch_pairs": dict(kind="pretraining", cfg=lambda: synth.tiny_config(in_batch_pairs=True), batch=3,
                                n_tok=7, n_reg=5),
    "tiny_roberta": dict(kind="vltasks", cfg=lambda: synth.tiny_config(model="roberta"), batch=2, n_tok=6, n_reg=4),
    # BASELINE.json configs[0]: bert_base_2layer_2conect.json, batch 8, 36 regions, 20 tokens
    "base_2l2c_b8": dict(kind="vltasks", cfg=lambda: synth.load_config("bert_base_2layer_2conect.json"),
                         batch=8, n_tok=20, n_reg=36,
                         stride={"vision_prediction": 16, "linguisic_prediction": 97}),
    # north-star model (bert_base_6layer_6conect.json) at the metric's sequence shape, small batch
    "base_6l6c_b2": dict(kind="vltasks", cfg=lambda: synth.load_config("bert_base_6layer_6conect.json"),
                         batch=2, n_tok=36, n_reg=36,
                         stride={"vision_prediction": 16, "linguisic_prediction": 97,
                                 "vil_prediction": 3, "vil_prediction_gqa": 3}),
    # BASELINE.json configs[2] shape: train_concap objective on the north-star model - 36 tokens, 36 regions + the
    # global mean-region row (R = 37), loader label conventions; the three losses of the REAL reference
    # (reference vilbert.py:1471-1597)
    "base_6l6c_concap_losses_b4": dict(kind="pretraining", cfg=lambda: synth.load_config("bert_base_6layer_6conect.json"),
                                       batch=4, n_tok=36, n_reg=37, labels=True),
    # the same model's reference-shaped score tensors (no labels) at R = 37
    "base_6l6c_concap_scores_b2": dict(kind="pretraining", cfg=lambda: synth.load_config("bert_base_6layer_6conect.json"),
                                       batch=2, n_tok=36, n_reg=37,
                                       stride={"prediction_scores_t": 97, "prediction_scores_v": 16}),
    # BASELINE.json configs[3] model: bert_large_6layer_6conect.json (H = 1024, 16 heads x 64, I = 4096, 24 text
    # layers, t_biattention_id = [18..23])
    "large_6l6c_b2": dict(kind="vltasks", cfg=lambda: synth.load_config("bert_large_6layer_6conect.json"),
                          batch=2, n_tok=36, n_reg=36,
                          stride={"vision_prediction": 16, "linguisic_prediction": 97,
                                  "vil_prediction": 3, "vil_prediction_gqa": 3}),
}


def case_inputs(case):
    c = CASES[case]
    cfg = c["cfg"]()
    x = synth.make_inputs(cfg, c["batch"], c["n_tok"], c["n_reg"], with_labels=c.get("labels", False),
                          task_id=c.get("task_id"))
    if c.get("text_batch"):  # one caption for the whole image batch
        tb = c["text_batch"]
        for k in ("input_ids", "token_type_ids", "attention_mask"):
            x[k] = x[k][:tb].contiguous()
        x["co_attention_mask"] = x["co_attention_mask"][:tb].contiguous()
    sd = synth.make_state_dict(cfg, c["kind"])
    return cfg, sd, x


def forward_args(case, x):
    """Positional argument tuple for the model's forward (the callers pass positionally)."""
    c = CASES[case]
    if c["kind"] == "vltasks":
        args = [x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
                x["image_attention_mask"], x["co_attention_mask"]]
        if "task_ids" in x:
            args.append(x["task_ids"])
        return tuple(args)
    args = [x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
            x["image_attention_mask"]]
    if c.get("labels"):
        args += [x["masked_lm_labels"], x["image_label"], x["image_target"], x["next_sentence_label"]]
    return tuple(args)


def output_names(case):
    c = CASES[case]
    if c["kind"] == "vltasks":
        return VL_NAMES
    return LOSS_NAMES if c.get("labels") else PT_NAMES


def sample(case, name, t):
    step = CASES[case].get("stride", {}).get(name, 1)
    return t[..., ::step] if step > 1 else t


def path(case):
    return os.path.join(GOLDEN_DIR, case + ".npz")
