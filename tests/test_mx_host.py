"""Host-side decisions of the MX inference mode (vilbert/ops.py), no GPU needed: which linears of the shipped configs run on
the MX GEMM, which heads are padded, when the bf16 attention / bf16 residual stream are asked for, and that none of it is
active under autograd or outside the mode."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))


@pytest.fixture
def mx():
    import __graft_entry__
    __graft_entry__.build()
    from vilbert import _native
    prev = _native.set_gemm_mode("mxfp8")
    yield
    _native.set_gemm_mode(prev)


def _cfg(name):
    return json.load(open(os.path.join(ROOT, "vilbert-multi-task_amd", "config", name)))


@pytest.mark.parametrize("name", ["bert_base_6layer_6conect.json", "bert_large_6layer_6conect.json"])
def test_every_encoder_linear_of_the_north_star_configs_is_mx_eligible(mx, name):
    from vilbert import ops
    c = _cfg(name)
    H, I, Hv, Iv, Hb = c["hidden_size"], c["intermediate_size"], c["v_hidden_size"], c["v_intermediate_size"], c["bi_hidden_size"]
    shapes = [(H, 3 * H), (H, H), (H, I), (I, H), (Hv, 3 * Hv), (Hv, Hv), (Hv, Iv), (Iv, Hv), (Hv, 3 * Hb), (H, 3 * Hb), (Hb, Hv),
              (Hb, H), (c["v_feature_size"], Hv)]
    with torch.no_grad():
        for K, N in shapes:
            assert ops.mx_eligible(K, N, None), (K, N)
        assert ops.mx_eligible(H, I, "gelu") and not ops.mx_eligible(H, Hb, "relu")          # poolers: ReLU -> row-scaled kernel
        assert ops.mx_attention_ok(36, 37, H // c["num_attention_heads"]) and ops.mx_attention_ok(37, 37, Hv // c["v_num_attention_heads"])
        # the task shapes (101 / 200 regions) stay on bf16 q | k | v too - on the key-tiled bf16 kernel; past MAX_KEYS nothing does
        assert ops.mx_attention_ok(23, 101, 128) and ops.mx_attention_ok(200, 20, 64) and not ops.mx_attention_ok(36, 36, 32)
        assert not ops.mx_attention_ok(36, ops.MAX_KEYS + 1, 64)
        assert not ops.mx_attention_ok(36, 36, 64, drop_p=0.1) and not ops.mx_attention_ok(36, 36, 64, other=True)
        assert ops.mx_stream_bf16()
    assert not ops.mx_eligible(H, H, None) and not ops.mx_attention_ok(36, 36, 64) and not ops.mx_stream_bf16()   # grad mode on


def test_ragged_heads_are_padded_and_tiny_heads_stay_fp32(mx):
    from vilbert import ops
    with torch.no_grad():
        for K, N in [(768, 30522), (1024, 1601), (2048, 3129), (2048, 1533)]:
            assert not ops.mx_eligible(K, N, None) and ops._mx_pad_eligible(K, N, None, 0.0, False, None, None, "f32"), (K, N)
        for K, N in [(1024, 2), (1024, 1), (1024, 3), (768, 100)]:
            assert not ops._mx_pad_eligible(K, N, None, 0.0, False, None, None, "f32"), (K, N)
        assert not ops._mx_pad_eligible(1024, 1601, None, 0.0, False, None, torch.zeros(1), "f32")     # no residual on a padded head
        assert not ops._mx_pad_eligible(1024, 1601, None, 0.0, False, None, None, "bf16")
        assert not ops._mx_pad_eligible(1000, 1601, None, 0.0, False, None, None, "f32")               # K must fit the K tile
        assert not ops.mx_eligible(768, 768, None, drop_p=0.1) and not ops.mx_eligible(768, 768, None, want_pre=True)


def test_mode_off_means_off():
    import __graft_entry__
    __graft_entry__.build()
    from vilbert import _native, ops
    assert _native.set_gemm_mode("f32") in ("f32", "bf16x6", "bf16x3", "bf16", "fp8", "fp8+bf16", "mxfp8")
    with torch.no_grad():
        assert not ops.mx_eligible(768, 768, None) and not ops.mx_attention_ok(36, 36, 64) and not ops.mx_stream_bf16()
    m = ops.MxRows.__new__(ops.MxRows)
    assert ops.MxRows.requires_grad is False and hasattr(ops.MxRows, "shape")
