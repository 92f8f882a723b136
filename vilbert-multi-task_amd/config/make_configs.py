"""Writes the model-config JSONs consumed by BertConfig.from_json_file.

Same file names, keys and values as the reference's config/*.json (hyper-parameters are facts:
reference config/bert_base_6layer_6conect.json:1-30 etc.), generated from one table so the
family stays consistent. Keys the reference never reads (bi_intermediate_size,
bi_attention_type, pooling_method, v_initializer_range) are kept so files are interchangeable.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
TEXT = {
    "base": dict(hidden_size=768, intermediate_size=3072, num_attention_heads=12, num_hidden_layers=12),
    "large": dict(hidden_size=1024, intermediate_size=4096, num_attention_heads=16, num_hidden_layers=24),
}
COMMON = dict(attention_probs_dropout_prob=0.1, hidden_act="gelu", hidden_dropout_prob=0.1,
              initializer_range=0.02, max_position_embeddings=512, type_vocab_size=2, vocab_size=30522)
VISION = dict(v_feature_size=2048, v_target_size=1601, v_hidden_size=1024, v_num_attention_heads=8,
              v_intermediate_size=1024, bi_hidden_size=1024, bi_num_attention_heads=8,
              bi_intermediate_size=1024, bi_attention_type=1, v_attention_probs_dropout_prob=0.1,
              v_hidden_act="gelu", v_hidden_dropout_prob=0.1, v_initializer_range=0.02,
              pooling_method="mul")


def two_stream(size, n):
    d = dict(COMMON, **TEXT[size])
    d.update(VISION, v_num_hidden_layers=n)
    L = d["num_hidden_layers"]
    d["v_biattention_id"] = list(range(n))
    d["t_biattention_id"] = list(range(L - n, L))
    return d


def main():
    out = {}
    for size in ("base", "large"):
        out["bert_%s_baseline.json" % size] = dict(COMMON, **TEXT[size])
        for n in (2, 4, 6) + ((8,) if size == "base" else ()):
            out["bert_%s_%dlayer_%dconect.json" % (size, n, n)] = two_stream(size, n)
    rb = two_stream("base", 6)
    rb.update(max_position_embeddings=514, type_vocab_size=1, vocab_size=50265, model="roberta",
              finetuning_task=None, layer_norm_eps=1e-12, num_labels=2, output_attentions=False,
              output_hidden_states=False, torchscript=False)
    out["roberta_base_6layer_6connect.json"] = rb
    for name, d in out.items():
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(d, f, indent=2)
            f.write("\n")


if __name__ == "__main__":
    main()
