"""ViLBERT two-stream model for MI355X - drop-in for the reference's ``vilbert.vilbert``.

Same public classes, constructor / forward signatures, config semantics and ``state_dict`` names
as /root/reference/vilbert/vilbert.py (cited per class), but every module's forward enqueues the
hand-written gfx950 kernels of libvilbert_hip.so instead of torch ops:

* q/k/v (and the 3+3 co-attention projections) run as one segmented GEMM per input stream;
* scale + mask + softmax + P.V + head merge is one attention kernel, no [B,h,S,S] tensor in HBM;
* bias, GELU / ReLU and the residual add are GEMM epilogues; LayerNorm is one row kernel;
* embeddings are gather + sum + LayerNorm in one kernel.

``nn.Linear`` / ``nn.Embedding`` / ``nn.Dropout`` children exist so that parameter names, shapes,
registration order and ``init_weights`` behave exactly like the reference; their own ``forward`` is
never used. There is no CPU path: calling a model on CPU tensors raises.
"""
import copy
import json
import logging
import os
import sys

import torch
from torch import nn
from torch.nn import CrossEntropyLoss
import torch.nn.functional as TF

from . import _native
from . import functional as F
from . import layers
from . import ops
from .utils import PreTrainedModel

logger = logging.getLogger(__name__)

_SUPPORTED_ACTS = ("gelu", "relu", "swish")


class BertConfig(object):
    """Reference vilbert.py:141-294: same constructor arguments, JSON semantics (``from_dict`` builds
    the defaults through ``BertConfig(-1)`` and then overwrites ``__dict__``, so keys the code never
    reads - ``bi_intermediate_size``, ``pooling_method`` ... - are tolerated and kept)."""

    def __init__(self, vocab_size_or_config_json_file, hidden_size=768, num_hidden_layers=12,
                 num_attention_heads=12, intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                 initializer_range=0.02, v_feature_size=2048, v_target_size=1601, v_hidden_size=768,
                 v_num_hidden_layers=3, v_num_attention_heads=12, v_intermediate_size=3072,
                 bi_hidden_size=1024, bi_num_attention_heads=16, v_attention_probs_dropout_prob=0.1,
                 v_hidden_act="gelu", v_hidden_dropout_prob=0.1, v_initializer_range=0.2,
                 v_biattention_id=[0, 1], t_biattention_id=[10, 11], visual_target=0, fast_mode=False,
                 fixed_v_layer=0, fixed_t_layer=0, in_batch_pairs=False, fusion_method="mul",
                 dynamic_attention=False, with_coattention=True, objective=0, num_negative=128,
                 model="bert", task_specific_tokens=False, visualization=False):
        assert len(v_biattention_id) == len(t_biattention_id)
        assert max(v_biattention_id) < v_num_hidden_layers
        assert max(t_biattention_id) < num_hidden_layers
        if isinstance(vocab_size_or_config_json_file, str):
            with open(vocab_size_or_config_json_file, "r", encoding="utf-8") as reader:
                self.__dict__.update(json.loads(reader.read()))
        elif isinstance(vocab_size_or_config_json_file, int):
            fields = dict(locals())
            fields.pop("self")
            self.vocab_size = fields.pop("vocab_size_or_config_json_file")
            self.__dict__.update(fields)
        else:
            raise ValueError("First argument must be either a vocabulary size (int)"
                             "or the path to a pretrained model config file (str)")

    @classmethod
    def from_dict(cls, json_object):
        config = BertConfig(vocab_size_or_config_json_file=-1)
        config.__dict__.update(json_object)
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding="utf-8") as reader:
            return cls.from_dict(json.loads(reader.read()))

    def __repr__(self):
        return str(self.to_json_string())

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"


# Two HIP streams: the text and the image stream of the encoder are independent between two connection
# layers (and so are the two halves of a connection layer around the bi-attention), so their kernels can
# share the chip - one stream's LayerNorm / attention / tail tiles run beside the other's GEMMs. Autograd
# replays every backward op on the stream its forward ran on, so the overlap carries over to backward.
# VB_TWO_STREAMS=0 turns it off.
_TWO_STREAMS = os.environ.get("VB_TWO_STREAMS", "1") != "0"
# inside a HIP-graph capture the fork / join becomes two parallel branches of the graph (VB_GRAPH_STREAMS=0: one chain)
_TWO_STREAMS_IN_GRAPH = os.environ.get("VB_GRAPH_STREAMS", "1") != "0"
_SIDE_STREAMS = {}


def set_two_streams(on):
    """Turn the text || image stream overlap on / off at run time; returns the previous setting. (bench.py
    serialises the step it brackets with per-launch HIP events, so kernel durations are not shared-chip times.)"""
    global _TWO_STREAMS
    prev, _TWO_STREAMS = _TWO_STREAMS, bool(on)
    return prev


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def _concurrent(side_fn, main_fn, side_inputs, enabled=True):
    """Run side_fn() on the side stream and main_fn() on the current stream, join, return both results.
    side_inputs: tensors (allocated on the current stream) the side work reads. Falls back to sequential
    execution (side first) when two-stream mode is off, on CPU tensors, or while a graph is being captured."""
    t0 = side_inputs[0]
    if not (enabled and _TWO_STREAMS and t0.is_cuda) or \
            (torch.cuda.is_current_stream_capturing() and not _TWO_STREAMS_IN_GRAPH):
        return side_fn(), main_fn()
    main = torch.cuda.current_stream(t0.device)
    side = _side_stream(t0.device)
    side.wait_stream(main)
    def _record(t, stream):
        if isinstance(t, ops.MxRows):                   # a result that exists only as MX codes + scale words
            t.q.record_stream(stream)
            t.s.record_stream(stream)
            return
        t.record_stream(stream)
        for extra in getattr(t, "_vb_fp8", ())[:2]:     # e4m3 codes + scales riding on a LayerNorm output (ops.py)
            extra.record_stream(stream)
        mx = getattr(t, "_vb_mx", None)                 # the same in the MX mode
        if mx is not None:
            mx[0].q.record_stream(stream)
            mx[0].s.record_stream(stream)

    for t in side_inputs:
        _record(t, side)            # allocated on `main`, consumed by kernels on `side`
    with torch.cuda.stream(side):
        a = side_fn()
    b = main_fn()
    main.wait_stream(side)
    for t in (a if isinstance(a, (tuple, list)) else (a,)):
        if torch.is_tensor(t) or isinstance(t, ops.MxRows):
            _record(t, main)        # allocated on `side`, consumed from here on by kernels on `main`
    return a, b


def _check_head_dim(d, what):
    """The native attention kernels are compiled for head dimensions 32, 64 and 128 (every shipped config: 64 / 128)
    and sequences up to 320 keys; say so at construction time instead of failing inside the first forward."""
    if d not in (32, 64, 128):
        raise NotImplementedError("%s: attention head size %d - the native attention kernels are compiled for "
                                  "head sizes 32, 64 and 128" % (what, d))


def _act_name(act):
    if not isinstance(act, str) or act not in _SUPPORTED_ACTS:
        raise NotImplementedError("activation %r: the native GEMM epilogues implement %s"
                                  % (act, ", ".join(_SUPPORTED_ACTS)))
    return act


def _drop_p(dropout_module):
    """Effective dropout probability of a reference-named nn.Dropout child (0 in eval mode)."""
    return dropout_module.p if dropout_module.training else 0.0


def _dropout(x, dropout_module):
    return F.dropout(x, _drop_p(dropout_module))


class BertLayerNorm(nn.Module):
    """TF-style LayerNorm (eps inside the sqrt), reference vilbert.py:304-317."""

    def __init__(self, hidden_size, eps=1e-12):
        super(BertLayerNorm, self).__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        return F.layer_norm(x, self.weight, self.bias, self.variance_epsilon)


class BertEmbeddings(nn.Module):
    """Reference vilbert.py:320-367."""

    def __init__(self, config):
        super(BertEmbeddings, self).__init__()
        self.task_specific_tokens = config.task_specific_tokens
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        if self.task_specific_tokens:
            self.task_embeddings = nn.Embedding(20, config.hidden_size)

    def forward(self, input_ids, token_type_ids=None, task_ids=None, position_ids=None):
        # position_ids is accepted and ignored exactly like the reference (:349-352 overwrite it)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        task_emb = None
        if self.task_specific_tokens:
            if task_ids is None:
                raise TypeError("task_specific_tokens=True needs task_ids")
            task_emb = self.task_embeddings.weight
        else:
            task_ids = None
        out = F.text_embed_ln(input_ids.long(), token_type_ids.long(), self.word_embeddings.weight,
                              self.position_embeddings.weight, self.token_type_embeddings.weight,
                              self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.variance_epsilon,
                              task_ids.long() if task_ids is not None else None, task_emb)
        return _dropout(out, self.dropout)


class RobertaEmbeddings(BertEmbeddings):
    """Reference vilbert.py:370-393. The reference computes RoBERTa position ids (padding_idx + 1 ...)
    and then its base class overwrites them with arange(seq_len) (:349-352), so the effective
    behaviour - kept here - is identical to BertEmbeddings; the third positional argument BertModel
    passes (task_ids) lands in ``position_ids`` and is dropped, as in the reference."""

    def __init__(self, config):
        super(RobertaEmbeddings, self).__init__(config)
        self.padding_idx = 1

    def forward(self, input_ids, token_type_ids=None, position_ids=None):
        return super(RobertaEmbeddings, self).forward(input_ids, token_type_ids=token_type_ids)


def _self_attention(mod, hidden_states, attention_mask, gates=None):
    """Shared body of BertSelfAttention / BertImageSelfAttention: fused q|k|v GEMM + attention kernel."""
    H = mod.all_head_size
    # MX inference mode: the projection leaves its GEMM as bf16 and the attention kernel returns the context as MX codes
    S = hidden_states.shape[1] if hidden_states.dim() == 3 else 10 ** 9
    # (decided for the whole block: the consuming output projection must be able to take an MX context - in train mode
    # under no_grad its hidden dropout makes it ineligible)
    to_mx = not mod.training and ops.mx_attention_ok(S, S, mod.attention_head_size, _drop_p(mod.dropout),
                                                     mod.visualization or gates is not None)
    qkv = F.linear(hidden_states, [mod.query.weight, mod.key.weight, mod.value.weight],
                   [mod.query.bias, mod.key.bias, mod.value.bias], out="bf16" if to_mx else "f32")
    if gates is not None:  # dynamic_attention (:577-586): rare path, small elementwise gates kept in torch
        qkv = torch.cat([qkv[..., :H] * gates[0].unsqueeze(1), qkv[..., H:2 * H] * gates[1].unsqueeze(1),
                         qkv[..., 2 * H:]], dim=-1)
    ctx, probs = F.self_attention(qkv, attention_mask, mod.num_attention_heads, _drop_p(mod.dropout),
                                  mod.visualization)
    attn_data = None
    if mod.visualization:
        B, S = qkv.shape[0], qkv.shape[1]
        split = lambda t: t.reshape(B, S, mod.num_attention_heads, mod.attention_head_size).permute(0, 2, 1, 3)
        attn_data = {"attn": probs, "queries": split(qkv[..., :H]), "keys": split(qkv[..., H:2 * H])}
    return ctx, attn_data


class BertSelfAttention(nn.Module):
    """Reference vilbert.py:396-460."""

    def __init__(self, config):
        super(BertSelfAttention, self).__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention "
                             "heads (%d)" % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        _check_head_dim(self.attention_head_size, "BertSelfAttention")
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.visualization = config.visualization
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def forward(self, hidden_states, attention_mask):
        return _self_attention(self, hidden_states, attention_mask)


def _dense_dropout_add_norm(dense, dropout, norm, hidden_states, input_tensor):
    """LayerNorm(dropout(dense(h)) + input): bias, dropout mask and residual add are all the GEMM epilogue
    (training and eval), followed by one LayerNorm pass."""
    # (MX inference mode: the sum leaves the GEMM as bf16 and the LayerNorm keeps the residual stream in bf16)
    return norm(F.linear(hidden_states, dense.weight, dense.bias, residual=input_tensor, drop_p=_drop_p(dropout),
                         out="bf16" if ops.mx_stream_bf16() and not dense.training else "f32"))


def _ffn(intermediate, output, x):
    """output(intermediate(x), x) = LayerNorm(dropout(dense2(act(dense1(x)))) + x) as one fused autograd node
    + the LayerNorm (``intermediate`` / ``output`` are the reference-named modules that own the parameters)."""
    y = F.ffn(x, intermediate.dense.weight, intermediate.dense.bias, intermediate.intermediate_act_fn,
              output.dense.weight, output.dense.bias, _drop_p(output.dropout))
    return output.LayerNorm(y)


class _ResidualNormOutput(nn.Module):
    """LayerNorm(dropout(dense(h)) + input): BertSelfOutput / BertOutput / BertImageSelfOutput /
    BertImageOutput (reference vilbert.py:463-474, 506-517, 622-633, 667-678)."""

    def __init__(self, in_features, out_features, dropout_prob):
        super(_ResidualNormOutput, self).__init__()
        self.dense = nn.Linear(in_features, out_features)
        self.LayerNorm = BertLayerNorm(out_features, eps=1e-12)
        self.dropout = nn.Dropout(dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return _dense_dropout_add_norm(self.dense, self.dropout, self.LayerNorm, hidden_states, input_tensor)


class BertSelfOutput(_ResidualNormOutput):
    def __init__(self, config):
        super(BertSelfOutput, self).__init__(config.hidden_size, config.hidden_size,
                                             config.hidden_dropout_prob)


class BertAttention(nn.Module):
    """Reference vilbert.py:477-486."""

    def __init__(self, config):
        super(BertAttention, self).__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask):
        self_output, attention_probs = self.self(input_tensor, attention_mask)
        return self.output(self_output, input_tensor), attention_probs


class BertIntermediate(nn.Module):
    """gelu(dense(h)) as one GEMM with a GELU epilogue. Reference vilbert.py:489-503."""

    def __init__(self, config):
        super(BertIntermediate, self).__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        self.intermediate_act_fn = _act_name(config.hidden_act)

    def forward(self, hidden_states):
        return F.linear(hidden_states, self.dense.weight, self.dense.bias, act=self.intermediate_act_fn)


class BertOutput(_ResidualNormOutput):
    def __init__(self, config):
        super(BertOutput, self).__init__(config.intermediate_size, config.hidden_size,
                                         config.hidden_dropout_prob)


class BertLayer(nn.Module):
    """Reference vilbert.py:520-533."""

    def __init__(self, config):
        super(BertLayer, self).__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask):
        # one autograd node / one call across the C ABI for the whole layer (layers.py, csrc/layers.hip) ...
        y = layers.self_layer(self, hidden_states, attention_mask, _drop_p(self.attention.self.dropout),
                              _drop_p(self.attention.output.dropout), _drop_p(self.output.dropout))
        if y is not None:
            return y, None
        # ... or op by op (attention maps wanted, fp8 / MX inference, shapes the launcher does not take)
        attention_output, attention_probs = self.attention(hidden_states, attention_mask)
        return _ffn(self.intermediate, self.output, attention_output), attention_probs


class BertImageSelfAttention(nn.Module):
    """Reference vilbert.py:536-619."""

    def __init__(self, config):
        super(BertImageSelfAttention, self).__init__()
        if config.v_hidden_size % config.v_num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention "
                             "heads (%d)" % (config.v_hidden_size, config.v_num_attention_heads))
        self.dynamic_attention = config.dynamic_attention
        self.num_attention_heads = config.v_num_attention_heads
        self.attention_head_size = int(config.v_hidden_size / config.v_num_attention_heads)
        _check_head_dim(self.attention_head_size, "BertImageSelfAttention")
        self.visualization = config.visualization
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.v_hidden_size, self.all_head_size)
        self.key = nn.Linear(config.v_hidden_size, self.all_head_size)
        self.value = nn.Linear(config.v_hidden_size, self.all_head_size)
        if self.dynamic_attention:
            self.dyLinear_q = nn.Linear(config.hidden_size, self.all_head_size)
            self.dyLinear_k = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.v_attention_probs_dropout_prob)

    def forward(self, hidden_states, attention_mask, txt_embedding, txt_attention_mask):
        gates = None
        if self.dynamic_attention:
            pool = (txt_embedding * txt_attention_mask).sum(1) / txt_attention_mask.sum(1)
            gates = (1 + torch.sigmoid(F.linear(pool, self.dyLinear_q.weight, self.dyLinear_q.bias)),
                     1 + torch.sigmoid(F.linear(pool, self.dyLinear_k.weight, self.dyLinear_k.bias)))
        return _self_attention(self, hidden_states, attention_mask, gates)


class BertImageSelfOutput(_ResidualNormOutput):
    def __init__(self, config):
        super(BertImageSelfOutput, self).__init__(config.v_hidden_size, config.v_hidden_size,
                                                  config.v_hidden_dropout_prob)


class BertImageAttention(nn.Module):
    """Reference vilbert.py:636-647."""

    def __init__(self, config):
        super(BertImageAttention, self).__init__()
        self.self = BertImageSelfAttention(config)
        self.output = BertImageSelfOutput(config)

    def forward(self, input_tensor, attention_mask, txt_embedding, txt_attention_mask):
        self_output, attention_probs = self.self(input_tensor, attention_mask, txt_embedding, txt_attention_mask)
        return self.output(self_output, input_tensor), attention_probs


class BertImageIntermediate(nn.Module):
    """Reference vilbert.py:650-664."""

    def __init__(self, config):
        super(BertImageIntermediate, self).__init__()
        self.dense = nn.Linear(config.v_hidden_size, config.v_intermediate_size)
        self.intermediate_act_fn = _act_name(config.v_hidden_act)

    def forward(self, hidden_states):
        return F.linear(hidden_states, self.dense.weight, self.dense.bias, act=self.intermediate_act_fn)


class BertImageOutput(_ResidualNormOutput):
    def __init__(self, config):
        super(BertImageOutput, self).__init__(config.v_intermediate_size, config.v_hidden_size,
                                              config.v_hidden_dropout_prob)


class BertImageLayer(nn.Module):
    """Reference vilbert.py:681-694."""

    def __init__(self, config):
        super(BertImageLayer, self).__init__()
        self.attention = BertImageAttention(config)
        self.intermediate = BertImageIntermediate(config)
        self.output = BertImageOutput(config)

    def forward(self, hidden_states, attention_mask, txt_embedding, txt_attention_mask):
        y = layers.self_layer(self, hidden_states, attention_mask, _drop_p(self.attention.self.dropout),
                              _drop_p(self.attention.output.dropout), _drop_p(self.output.dropout))
        if y is not None:
            return y, None
        attention_output, attention_probs = self.attention(hidden_states, attention_mask, txt_embedding,
                                                           txt_attention_mask)
        return _ffn(self.intermediate, self.output, attention_output), attention_probs


class BertBiAttention(nn.Module):
    """Bi-directional co-attention, reference vilbert.py:697-823. Stream 1 = image regions, stream 2 =
    text tokens. ``co_attention_mask`` is accepted and ignored, as in the reference (:774-775,796-797)."""

    def __init__(self, config):
        super(BertBiAttention, self).__init__()
        if config.bi_hidden_size % config.bi_num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention "
                             "heads (%d)" % (config.bi_hidden_size, config.bi_num_attention_heads))
        self.visualization = config.visualization
        self.num_attention_heads = config.bi_num_attention_heads
        self.attention_head_size = int(config.bi_hidden_size / config.bi_num_attention_heads)
        _check_head_dim(self.attention_head_size, "BertBiAttention")
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query1 = nn.Linear(config.v_hidden_size, self.all_head_size)
        self.key1 = nn.Linear(config.v_hidden_size, self.all_head_size)
        self.value1 = nn.Linear(config.v_hidden_size, self.all_head_size)
        self.dropout1 = nn.Dropout(config.v_attention_probs_dropout_prob)
        self.query2 = nn.Linear(config.hidden_size, self.all_head_size)
        self.key2 = nn.Linear(config.hidden_size, self.all_head_size)
        self.value2 = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout2 = nn.Dropout(config.attention_probs_dropout_prob)

    def forward(self, input_tensor1, attention_mask1, input_tensor2, attention_mask2, co_attention_mask=None,
                use_co_attention_mask=False):
        H = self.all_head_size
        S1, S2 = input_tensor1.shape[1], input_tensor2.shape[1]
        to_mx = (not self.training and ops.mx_attention_ok(max(S1, S2), max(S1, S2), self.attention_head_size,
                                     max(_drop_p(self.dropout1), _drop_p(self.dropout2)), self.visualization)
                 and all(ops.mx_eligible(w.shape[1], 3 * w.shape[0]) for w in (self.query1.weight, self.query2.weight)))
        fmt = "bf16" if to_mx else "f32"
        qkv1, qkv2 = _concurrent(   # the image-side and text-side projections are independent
            lambda: F.linear(input_tensor1, [self.query1.weight, self.key1.weight, self.value1.weight],
                             [self.query1.bias, self.key1.bias, self.value1.bias], out=fmt),
            lambda: F.linear(input_tensor2, [self.query2.weight, self.key2.weight, self.value2.weight],
                             [self.query2.bias, self.key2.bias, self.value2.bias], out=fmt),
            [input_tensor1])
        # context_layer1: text queries over image keys / values -> TEXT stream (:768-785, dropout1)
        # context_layer2: image queries over text keys / values -> IMAGE stream (:787-809, dropout2)
        context_layer1, context_layer2, probs1, probs2 = F.bi_attention(
            qkv1, qkv2, attention_mask1, attention_mask2, self.num_attention_heads, _drop_p(self.dropout1),
            _drop_p(self.dropout2), self.visualization)
        attn_data = None
        if self.visualization:
            nh, hd = self.num_attention_heads, self.attention_head_size
            split = lambda t: t.reshape(t.shape[0], t.shape[1], nh, hd).permute(0, 2, 1, 3)
            attn_data = {"attn1": probs1, "queries1": split(qkv2[..., :H]), "keys1": split(qkv1[..., H:2 * H]),
                         "attn2": probs2, "querues2": split(qkv1[..., :H]), "keys2": split(qkv2[..., H:2 * H])}
        return context_layer1, context_layer2, attn_data


class BertBiOutput(nn.Module):
    """Reference vilbert.py:826-855. ``q_dense1/q_dense2`` (+ dropouts) are registered because the
    checkpoints carry them; no forward uses them (they never receive gradients)."""

    def __init__(self, config):
        super(BertBiOutput, self).__init__()
        self.dense1 = nn.Linear(config.bi_hidden_size, config.v_hidden_size)
        self.LayerNorm1 = BertLayerNorm(config.v_hidden_size, eps=1e-12)
        self.dropout1 = nn.Dropout(config.v_hidden_dropout_prob)
        self.q_dense1 = nn.Linear(config.bi_hidden_size, config.v_hidden_size)
        self.q_dropout1 = nn.Dropout(config.v_hidden_dropout_prob)
        self.dense2 = nn.Linear(config.bi_hidden_size, config.hidden_size)
        self.LayerNorm2 = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout2 = nn.Dropout(config.hidden_dropout_prob)
        self.q_dense2 = nn.Linear(config.bi_hidden_size, config.hidden_size)
        self.q_dropout2 = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states1, input_tensor1, hidden_states2, input_tensor2):
        out1 = _dense_dropout_add_norm(self.dense1, self.dropout1, self.LayerNorm1, hidden_states1, input_tensor1)
        out2 = _dense_dropout_add_norm(self.dense2, self.dropout2, self.LayerNorm2, hidden_states2, input_tensor2)
        return out1, out2


class BertConnectionLayer(nn.Module):
    """Reference vilbert.py:858-900."""

    def __init__(self, config):
        super(BertConnectionLayer, self).__init__()
        self.biattention = BertBiAttention(config)
        self.biOutput = BertBiOutput(config)
        self.v_intermediate = BertImageIntermediate(config)
        self.v_output = BertImageOutput(config)
        self.t_intermediate = BertIntermediate(config)
        self.t_output = BertOutput(config)

    def forward(self, input_tensor1, attention_mask1, input_tensor2, attention_mask2, co_attention_mask=None,
                use_co_attention_mask=False):
        bi, bo_ = self.biattention, self.biOutput
        out = layers.connection_layer(self, input_tensor1, attention_mask1, input_tensor2, attention_mask2,
                                      (_drop_p(bi.dropout1), _drop_p(bi.dropout2), _drop_p(bo_.dropout1),
                                       _drop_p(self.v_output.dropout), _drop_p(bo_.dropout2), _drop_p(self.t_output.dropout)),
                                      _concurrent)
        if out is not None:
            return out[0], out[1], None
        bi_output1, bi_output2, co_attention_probs = self.biattention(
            input_tensor1, attention_mask1, input_tensor2, attention_mask2, co_attention_mask,
            use_co_attention_mask)
        # cross-wiring of the reference call site (:890-892): the image stream takes the context
        # computed from image queries (bi_output2), the text stream the one from text queries.
        bo = self.biOutput

        def image_branch():
            a1 = _dense_dropout_add_norm(bo.dense1, bo.dropout1, bo.LayerNorm1, bi_output2, input_tensor1)
            return _ffn(self.v_intermediate, self.v_output, a1)

        def text_branch():
            a2 = _dense_dropout_add_norm(bo.dense2, bo.dropout2, bo.LayerNorm2, bi_output1, input_tensor2)
            return _ffn(self.t_intermediate, self.t_output, a2)

        layer_output1, layer_output2 = _concurrent(image_branch, text_branch, [bi_output2, input_tensor1])
        return layer_output1, layer_output2, co_attention_probs


class BertEncoder(nn.Module):
    """Layer schedule of the two streams, reference vilbert.py:903-1107."""

    def __init__(self, config):
        super(BertEncoder, self).__init__()
        self.FAST_MODE = config.fast_mode
        self.with_coattention = config.with_coattention
        self.v_biattention_id = config.v_biattention_id
        self.t_biattention_id = config.t_biattention_id
        self.in_batch_pairs = config.in_batch_pairs
        self.fixed_t_layer = config.fixed_t_layer
        self.fixed_v_layer = config.fixed_v_layer
        # the reference builds ONE layer of each kind and deep-copies it, so all text layers (and
        # all image / connection layers) start from identical weights before init_weights runs
        layer, v_layer, connect_layer = BertLayer(config), BertImageLayer(config), BertConnectionLayer(config)
        self.layer = nn.ModuleList([copy.deepcopy(layer) for _ in range(config.num_hidden_layers)])
        self.v_layer = nn.ModuleList([copy.deepcopy(v_layer) for _ in range(config.v_num_hidden_layers)])
        self.c_layer = nn.ModuleList([copy.deepcopy(connect_layer) for _ in range(len(config.v_biattention_id))])

    def forward(self, txt_embedding, image_embedding, txt_attention_mask, txt_attention_mask2,
                image_attention_mask, co_attention_mask=None, output_all_encoded_layers=True,
                output_all_attention_masks=False):
        v_start = t_start = count = 0
        all_encoder_layers_t, all_encoder_layers_v = [], []
        all_attention_mask_t, all_attnetion_mask_v, all_attention_mask_c = [], [], []
        batch_size, num_words, t_hidden_size = txt_embedding.size()
        _, num_regions, v_hidden_size = image_embedding.size()
        use_co_attention_mask = False
        dynamic = len(self.v_layer) > 0 and self.v_layer[0].attention.self.dynamic_attention

        def thaw(x, frozen):
            # a frozen prefix runs under no_grad: in the MX inference mode it may hand over a bf16 hidden state, which
            # the grad-enabled layers behind it (fp32 kernels unless the bf16 stream is on) cannot take
            if frozen and torch.is_grad_enabled() and x.dtype == torch.bfloat16 and not _native.bf16_stream():
                return F.to_f32(x)
            return x

        def run_text(lo, hi, x, frozen=False):
            for idx in range(lo, hi):
                with torch.set_grad_enabled(torch.is_grad_enabled() and not frozen):
                    x, probs = self.layer[idx](x, txt_attention_mask)
                if output_all_attention_masks:
                    all_attention_mask_t.append(probs)
            return thaw(x, frozen)

        def run_image(lo, hi, x, frozen=False):
            for idx in range(lo, hi):
                with torch.set_grad_enabled(torch.is_grad_enabled() and not frozen):
                    x, probs = self.v_layer[idx](x, image_attention_mask, txt_embedding, txt_attention_mask2)
                if output_all_attention_masks:
                    all_attnetion_mask_v.append(probs)
            return thaw(x, frozen)

        for v_end, t_end in zip(self.v_biattention_id, self.t_biattention_id):
            assert self.fixed_t_layer <= t_end
            assert self.fixed_v_layer <= v_end
            def text_part(x=txt_embedding, lo=t_start, hi=t_end):
                if lo < self.fixed_t_layer:
                    x = run_text(lo, self.fixed_t_layer, x, frozen=True)
                    lo = self.fixed_t_layer
                return run_text(lo, hi, x)

            def image_part(x=image_embedding, lo=v_start, hi=v_end):
                if lo < self.fixed_v_layer:
                    x = run_image(lo, self.fixed_v_layer, x, frozen=True)
                    lo = self.fixed_v_layer
                return run_image(lo, hi, x)

            if t_end > t_start and v_end > v_start and not dynamic:
                # independent stretches of the two streams: image layers on the side stream
                # (attention maps collected by the layers would escape the side stream: one stream then)
                image_embedding, txt_embedding = _concurrent(image_part, text_part,
                                                             [image_embedding, image_attention_mask],
                                                             enabled=not output_all_attention_masks)
            else:   # (dynamic attention makes the image layers read the UPDATED text stream, :577-586)
                txt_embedding = text_part()
                image_embedding = image_part()
            t_start, v_start = max(t_start, self.fixed_t_layer), max(v_start, self.fixed_v_layer)

            if count == 0 and self.in_batch_pairs:
                # every caption against every image: batch becomes batch_size ** 2 (:1008-1040)
                b = batch_size
                image_embedding = image_embedding.unsqueeze(0).expand(b, b, num_regions, v_hidden_size) \
                    .contiguous().view(b * b, num_regions, v_hidden_size)
                image_attention_mask = image_attention_mask.unsqueeze(0).expand(b, b, 1, 1, num_regions) \
                    .contiguous().view(b * b, 1, 1, num_regions)
                txt_embedding = txt_embedding.unsqueeze(1).expand(b, b, num_words, t_hidden_size) \
                    .contiguous().view(b * b, num_words, t_hidden_size)
                txt_attention_mask = txt_attention_mask.unsqueeze(1).expand(b, b, 1, 1, num_words) \
                    .contiguous().view(b * b, 1, 1, num_words)
                co_attention_mask = co_attention_mask.unsqueeze(1).expand(b, b, 1, num_regions, num_words) \
                    .contiguous().view(b * b, 1, num_regions, num_words)

            if count == 0 and self.FAST_MODE:
                # one caption against many images (:1042-1053)
                n = image_embedding.size(0)
                txt_embedding = txt_embedding.expand(n, txt_embedding.size(1), txt_embedding.size(2))
                txt_attention_mask = txt_attention_mask.expand(n, txt_attention_mask.size(1),
                                                               txt_attention_mask.size(2),
                                                               txt_attention_mask.size(3))

            if self.with_coattention:
                image_embedding, txt_embedding, co_attention_probs = self.c_layer[count](
                    image_embedding, image_attention_mask, txt_embedding, txt_attention_mask,
                    co_attention_mask, use_co_attention_mask)
                if output_all_attention_masks:
                    all_attention_mask_c.append(co_attention_probs)

            v_start, t_start = v_end, t_end
            count += 1
            if output_all_encoded_layers:
                all_encoder_layers_t.append(txt_embedding)
                all_encoder_layers_v.append(image_embedding)

        if len(self.v_layer) > v_start and len(self.layer) > t_start and not dynamic:
            image_embedding, txt_embedding = _concurrent(
                lambda: run_image(v_start, len(self.v_layer), image_embedding),
                lambda: run_text(t_start, len(self.layer), txt_embedding), [image_embedding, image_attention_mask],
                enabled=not output_all_attention_masks)
        else:
            image_embedding = run_image(v_start, len(self.v_layer), image_embedding)
            txt_embedding = run_text(t_start, len(self.layer), txt_embedding)

        if not output_all_encoded_layers:
            all_encoder_layers_t.append(txt_embedding)
            all_encoder_layers_v.append(image_embedding)
        return (all_encoder_layers_t, all_encoder_layers_v,
                (all_attention_mask_t, all_attnetion_mask_v, all_attention_mask_c))


class _FirstTokenPooler(nn.Module):
    """ReLU(dense(h[:, 0])) - the first-token rows are read in place through the GEMM's row stride
    and ReLU is its epilogue. Reference vilbert.py:1110-1137 (ReLU, not tanh)."""

    def __init__(self, in_features, out_features):
        super(_FirstTokenPooler, self).__init__()
        self.dense = nn.Linear(in_features, out_features)
        self.activation = nn.ReLU()

    def forward(self, hidden_states):
        return F.linear(hidden_states[:, 0], self.dense.weight, self.dense.bias, act="relu")


class BertTextPooler(_FirstTokenPooler):
    def __init__(self, config):
        super(BertTextPooler, self).__init__(config.hidden_size, config.bi_hidden_size)


class BertImagePooler(_FirstTokenPooler):
    def __init__(self, config):
        super(BertImagePooler, self).__init__(config.v_hidden_size, config.bi_hidden_size)


class BertPredictionHeadTransform(nn.Module):
    """Reference vilbert.py:1140-1156."""

    def __init__(self, config):
        super(BertPredictionHeadTransform, self).__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.transform_act_fn = _act_name(config.hidden_act)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)

    def forward(self, hidden_states):
        return self.LayerNorm(F.linear(hidden_states, self.dense.weight, self.dense.bias,
                                       act=self.transform_act_fn))


class BertImgPredictionHeadTransform(nn.Module):
    """Reference vilbert.py:1159-1175 - note it activates with ``hidden_act`` (the TEXT activation)
    whenever that is a string, which it always is for JSON configs."""

    def __init__(self, config):
        super(BertImgPredictionHeadTransform, self).__init__()
        self.dense = nn.Linear(config.v_hidden_size, config.v_hidden_size)
        self.transform_act_fn = _act_name(config.hidden_act if isinstance(config.hidden_act, str)
                                          else config.v_hidden_act)
        self.LayerNorm = BertLayerNorm(config.v_hidden_size, eps=1e-12)

    def forward(self, hidden_states):
        return self.LayerNorm(F.linear(hidden_states, self.dense.weight, self.dense.bias,
                                       act=self.transform_act_fn))


class BertLMPredictionHead(nn.Module):
    """Reference vilbert.py:1178-1196: decoder weight tied to the word embeddings + output-only bias."""

    def __init__(self, config, bert_model_embedding_weights):
        super(BertLMPredictionHead, self).__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(bert_model_embedding_weights.size(1), bert_model_embedding_weights.size(0),
                                 bias=False)
        self.decoder.weight = bert_model_embedding_weights
        self.bias = nn.Parameter(torch.zeros(bert_model_embedding_weights.size(0)))

    def forward(self, hidden_states, pad_cols=False):
        # pad_cols (internal, labelled-rows loss path): the [rows, 30522] logits live in a buffer with a 30524-float
        # row stride, so the rows of the logits and of their gradient are 16-byte aligned for the backward GEMMs
        return F.linear(self.transform(hidden_states), self.decoder.weight, self.bias, pad_cols=pad_cols)


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config, bert_model_embedding_weights):
        super(BertOnlyMLMHead, self).__init__()
        self.predictions = BertLMPredictionHead(config, bert_model_embedding_weights)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


class BertOnlyNSPHead(nn.Module):
    def __init__(self, config):
        super(BertOnlyNSPHead, self).__init__()
        self.seq_relationship = nn.Linear(config.hidden_size, 2)

    def forward(self, pooled_output):
        return F.linear(pooled_output, self.seq_relationship.weight, self.seq_relationship.bias)


class BertImagePredictionHead(nn.Module):
    """Reference vilbert.py:1246-1258."""

    def __init__(self, config):
        super(BertImagePredictionHead, self).__init__()
        self.transform = BertImgPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.v_hidden_size, config.v_target_size)

    def forward(self, hidden_states):
        return F.linear(self.transform(hidden_states), self.decoder.weight, self.decoder.bias)


def _capacity(positions, frac):
    """Fixed gather capacity: `frac` of the positions, a multiple of 32 (aligned wgrad contraction, whole MFMA tiles)."""
    return max(32, min((positions + 31) // 32 * 32, (int(positions * frac) + 31) // 32 * 32))


def _fuse_pooled(fusion_method, pooled_output_t, pooled_output_v):
    if fusion_method == "sum":
        return pooled_output_t + pooled_output_v
    if fusion_method == "mul":
        return pooled_output_t * pooled_output_v
    assert False


class BertPreTrainingHeads(nn.Module):
    """Reference vilbert.py:1219-1243."""

    def __init__(self, config, bert_model_embedding_weights):
        super(BertPreTrainingHeads, self).__init__()
        self.predictions = BertLMPredictionHead(config, bert_model_embedding_weights)
        self.bi_seq_relationship = nn.Linear(config.bi_hidden_size, 2)
        self.imagePredictions = BertImagePredictionHead(config)
        self.fusion_method = config.fusion_method
        self.dropout = nn.Dropout(0.1)

    def forward(self, sequence_output_t, sequence_output_v, pooled_output_t, pooled_output_v):
        pooled_output = _dropout(_fuse_pooled(self.fusion_method, pooled_output_t, pooled_output_v), self.dropout)
        prediction_scores_t = self.predictions(sequence_output_t)
        seq_relationship_score = F.linear(pooled_output, self.bi_seq_relationship.weight,
                                          self.bi_seq_relationship.bias)
        prediction_scores_v = self.imagePredictions(sequence_output_v)
        return prediction_scores_t, prediction_scores_v, seq_relationship_score


class BertPreTrainedModel(PreTrainedModel):
    """Weight initialisation + from_pretrained plumbing, reference vilbert.py:1261-1285."""

    config_class = BertConfig
    pretrained_model_archive_map = {}
    base_model_prefix = "bert"

    def __init__(self, *inputs, **kwargs):
        super(BertPreTrainedModel, self).__init__(*inputs, **kwargs)

    def half(self):
        """What the reference's scripts call for reduced-precision training (`model.half()` next to apex's FP16_Optimizer,
        which keeps fp32 master weights: train_concap.py:443-461,504-505; train_tasks.py:168-171). Here that mode is the bf16
        stream of DESIGN.md section 4.5: activations, saved tensors and activation gradients become bfloat16, the parameters
        STAY fp32 (they are the master weights; the kernels keep their own bf16 copies). The switch belongs to THIS model
        (round 6): its forward runs under `_native.set_gemm_mode("bf16")` and restores the process-wide mode afterwards, so a
        second model in the process (an evaluation copy, an fp8 / MX inference model) keeps its own arithmetic; `float()`
        undoes it. Backward needs no mode: the autograd nodes the bf16 forward recorded call the bf16 kernels directly."""
        self._vb_bf16 = True
        logger.info("half(): bf16 mode on for this model (bfloat16 activations / gradients, fp32 master weights); "
                    "parameters unchanged")
        return self

    def float(self):
        self._vb_bf16 = False
        return super(BertPreTrainedModel, self).float()

    def __call__(self, *args, **kwargs):
        if getattr(self, "_vb_bf16", False) and not _native.bf16_stream():
            prev = _native.set_gemm_mode("bf16")
            try:
                return super(BertPreTrainedModel, self).__call__(*args, **kwargs)
            finally:
                _native.set_gemm_mode(prev)
        return super(BertPreTrainedModel, self).__call__(*args, **kwargs)

    def init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()


class BertImageEmbeddings(nn.Module):
    """LayerNorm(Linear(2048->Hv)(feat) + Linear(5->Hv)(loc)), reference vilbert.py:1409-1432: the big
    projection is the MFMA GEMM (coalesced 128-byte reads of the [regions x 2048] features); the
    5-wide location projection, the sum and the LayerNorm are one row kernel."""

    def __init__(self, config):
        super(BertImageEmbeddings, self).__init__()
        self.image_embeddings = nn.Linear(config.v_feature_size, config.v_hidden_size)
        self.image_location_embeddings = nn.Linear(5, config.v_hidden_size)
        self.LayerNorm = BertLayerNorm(config.v_hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, input_loc):
        if _native.bf16_stream() and input_ids.is_cuda:
            # bf16 mode: the [regions x 2048] features are rounded to bf16 once (they need no gradient), the projection
            # runs on the bf16 GEMM with an fp32 result for the (fp32) location + sum + LayerNorm row kernel
            proj = F.linear_f32_out(F.to_bf16(input_ids.float()), self.image_embeddings.weight, self.image_embeddings.bias)
        else:
            proj = F.linear(input_ids.float(), self.image_embeddings.weight, self.image_embeddings.bias)
        out = F.image_embed_ln(proj, input_loc.float(), self.image_location_embeddings.weight,
                               self.image_location_embeddings.bias, self.LayerNorm.weight, self.LayerNorm.bias,
                               self.LayerNorm.variance_epsilon)
        return _dropout(out, self.dropout)


class BertModel(BertPreTrainedModel):
    """Reference vilbert.py:1288-1406."""

    def __init__(self, config):
        super(BertModel, self).__init__(config)
        if config.model == "bert":
            self.embeddings = BertEmbeddings(config)
        elif config.model == "roberta":
            self.embeddings = RobertaEmbeddings(config)
        self.task_specific_tokens = config.task_specific_tokens
        self.v_embeddings = BertImageEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.t_pooler = BertTextPooler(config)
        self.v_pooler = BertImagePooler(config)
        self.apply(self.init_weights)

    def forward(self, input_txt, input_imgs, image_loc, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, co_attention_mask=None, task_ids=None,
                output_all_encoded_layers=False, output_all_attention_masks=False):
        from . import ops
        if attention_mask is None:
            attention_mask = torch.ones_like(input_txt)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_txt)
        if image_attention_mask is None:
            image_attention_mask = torch.ones(input_imgs.size(0), input_imgs.size(1), dtype=input_txt.dtype,
                                              device=input_txt.device)      # (created on the device: graph-capturable)
        if self.task_specific_tokens:
            # the mask grows at position 0 (:1331-1334) while the task embedding sits at position 1
            mask_tokens = input_txt.new().resize_(input_txt.size(0), 1).fill_(1)
            attention_mask = torch.cat([mask_tokens.to(attention_mask.dtype), attention_mask], dim=1)

        # additive masks (1 - m) * -10000 in the [B,1,1,S] shape the encoder modules take (:1341-1362)
        extended_attention_mask = ops.additive_mask(attention_mask).unsqueeze(1).unsqueeze(2)
        extended_image_attention_mask = ops.additive_mask(image_attention_mask).unsqueeze(1).unsqueeze(2)
        extended_attention_mask2 = attention_mask.unsqueeze(2).to(dtype=next(self.parameters()).dtype)

        if co_attention_mask is None:
            co_attention_mask = torch.zeros(input_txt.size(0), input_imgs.size(1), input_txt.size(1),
                                            dtype=extended_image_attention_mask.dtype,
                                            device=extended_image_attention_mask.device)
        # scaled by 5 and then never consumed, as in the reference (:1364-1375)
        extended_co_attention_mask = (co_attention_mask.unsqueeze(1) * 5.0).to(
            dtype=next(self.parameters()).dtype)

        if _native.bf16_stream() and input_imgs.is_cuda:
            # the one-launch refresh of every bf16 weight shadow runs HERE, on the stream the text / image branches fork from
            # (advisor finding of round 5: left to the first linear that asks, it could run inside one branch while the
            # other already reads shadows)
            from . import ops16
            ops16.refresh_stale(input_imgs.device)
        embedding_output = self.embeddings(input_txt, token_type_ids, task_ids)
        v_embedding_output = self.v_embeddings(input_imgs, image_loc)
        if _native.bf16_stream() and embedding_output.is_cuda and not self.config.dynamic_attention:
            # bf16 mode (round 5): the two hidden-state streams enter the encoder as bfloat16 and stay bfloat16 - activations,
            # saved tensors and their gradients - until the casts below (functional.py dispatches on the dtype)
            embedding_output, v_embedding_output = F.to_bf16(embedding_output), F.to_bf16(v_embedding_output)
        encoded_layers_t, encoded_layers_v, all_attention_mask = self.encoder(
            embedding_output, v_embedding_output, extended_attention_mask, extended_attention_mask2,
            extended_image_attention_mask, extended_co_attention_mask,
            output_all_encoded_layers=output_all_encoded_layers,
            output_all_attention_masks=output_all_attention_masks)

        # bf16 mode / MX inference mode: the encoder keeps its residual stream in bf16; the callers (poolers, heads, users)
        # get fp32
        encoded_layers_t = [F.to_f32(t) for t in encoded_layers_t]
        encoded_layers_v = [F.to_f32(t) for t in encoded_layers_v]
        sequence_output_t, sequence_output_v = encoded_layers_t[-1], encoded_layers_v[-1]
        pooled_output_t = self.t_pooler(sequence_output_t)
        pooled_output_v = self.v_pooler(sequence_output_v)
        if not output_all_encoded_layers:
            encoded_layers_t, encoded_layers_v = encoded_layers_t[-1], encoded_layers_v[-1]
        return encoded_layers_t, encoded_layers_v, pooled_output_t, pooled_output_v, all_attention_mask


class BertForMultiModalPreTraining(BertPreTrainedModel):
    """Masked-LM + masked-region + alignment pre-training wrapper, reference vilbert.py:1435-1597."""

    def __init__(self, config):
        super(BertForMultiModalPreTraining, self).__init__(config)
        self.bert = BertModel(config)
        self.cls = BertPreTrainingHeads(config, self.bert.embeddings.word_embeddings.weight)
        self.apply(self.init_weights)
        self.visual_target = config.visual_target
        self.num_negative = config.num_negative
        # None: the labelled rows are gathered exactly (one host sync per step); a fraction in (0, 1]: sync-free
        # fixed-capacity gather of that share of the token / region positions (see _losses_at_labelled_positions)
        # None: exact gather of the labelled rows (one host sync per step); a fraction of the positions or "auto": sync-free
        # fixed-capacity gather (see _losses_at_labelled_positions)
        self.label_capacity = None
        self._auto_capacity = None
        self._label_counts = None
        self._label_overflow = None     # device flag: a fixed-capacity label gather dropped rows (sticky until checked)
        self.loss_fct = CrossEntropyLoss(ignore_index=-1)
        print("model's visual target is ", config.visual_target)
        if self.visual_target == 0:
            self.vis_criterion = nn.KLDivLoss(reduction="none")
        elif self.visual_target == 1:
            self.vis_criterion = nn.MSELoss(reduction="none")
        elif self.visual_target == 2:
            self.vis_criterion = CrossEntropyLoss()
        self.tie_weights()

    def tie_weights(self):
        self._tie_or_clone_weights(self.cls.predictions.decoder, self.bert.embeddings.word_embeddings)

    def forward(self, input_ids, image_feat, image_loc, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, masked_lm_labels=None, image_label=None, image_target=None,
                next_sentence_label=None, output_all_attention_masks=False):
        sequence_output_t, sequence_output_v, pooled_output_t, pooled_output_v, all_attention_mask = self.bert(
            input_ids, image_feat, image_loc, token_type_ids, attention_mask, image_attention_mask,
            output_all_encoded_layers=False, output_all_attention_masks=output_all_attention_masks)
        with_labels = not (masked_lm_labels is None or next_sentence_label is None or image_target is None)
        if with_labels and self.visual_target in (0, 1):
            losses = self._losses_at_labelled_positions(sequence_output_t, sequence_output_v, pooled_output_t,
                                                        pooled_output_v, masked_lm_labels, image_label,
                                                        image_target, next_sentence_label)
            if losses is not None:
                return losses
        prediction_scores_t, prediction_scores_v, seq_relationship_score = self.cls(
            sequence_output_t, sequence_output_v, pooled_output_t, pooled_output_v)
        if not with_labels:
            return prediction_scores_t, prediction_scores_v, seq_relationship_score, all_attention_mask

        prediction_scores_v = prediction_scores_v[:, 1:]
        labelled = image_label == 1
        if self.visual_target == 1:
            img_loss = self.vis_criterion(prediction_scores_v, image_target)
            masked_img_loss = torch.sum(img_loss * labelled.unsqueeze(2).float()) / max(
                torch.sum(labelled.unsqueeze(2).expand_as(img_loss)), 1)
        elif self.visual_target == 0:
            img_loss = self.vis_criterion(TF.log_softmax(prediction_scores_v, dim=2), image_target)
            # divisor quirk kept from the reference (:1520-1522): max(count, 0) -> NaN when no region
            # is labelled
            masked_img_loss = torch.sum(img_loss * labelled.unsqueeze(2).float()) / max(torch.sum(labelled), 0)
        elif self.visual_target == 2:
            masked_img_loss = self._nce_region_loss(input_ids, prediction_scores_v, image_target, labelled)
        masked_lm_loss = F.cross_entropy(prediction_scores_t.view(-1, self.config.vocab_size),
                                         masked_lm_labels.reshape(-1), ignore_index=-1)
        next_sentence_loss = F.cross_entropy(seq_relationship_score.view(-1, 2), next_sentence_label.reshape(-1),
                                             ignore_index=-1)
        return masked_lm_loss.unsqueeze(0), masked_img_loss.unsqueeze(0), next_sentence_loss.unsqueeze(0)

    def _losses_at_labelled_positions(self, sequence_output_t, sequence_output_v, pooled_output_t, pooled_output_v,
                                      masked_lm_labels, image_label, image_target, next_sentence_label):
        """The three pre-training losses with the heads evaluated ONLY where a label exists (~15 % of the
        tokens / regions): CrossEntropy(ignore_index=-1) ignores every other token row and the region loss
        multiplies every other region row by zero (reference vilbert.py:1506-1522,1578-1585), so the value
        and every gradient are the same while the [B,T,30522] logits tensor (1.1 GB at B=256) and 85 % of
        the two decoder GEMMs never exist. Costs one host sync for the row counts (the reference's training
        loop syncs every step anyway, train_concap.py:589-598). Returns None when nothing is labelled (the
        caller then takes the reference-shaped path, which yields the reference's NaN)."""
        cls = self.cls
        lm_flat = masked_lm_labels.reshape(-1)
        labelled = image_label == 1
        n_reg_all = sequence_output_v.size(1)                       # regions incl. the global row 0
        per = n_reg_all - 1
        static = (self.label_capacity is not None or torch.cuda.is_current_stream_capturing()) and self.visual_target == 0
        if static and self.label_capacity == "auto" and self._auto_capacity is None and not torch.cuda.is_current_stream_capturing():
            # "auto": the FIRST step counts its labelled rows on the host (one sync, the exact path below) and fixes the gather
            # capacity at 1.2 x the larger of the two labelled fractions (+ 1 % of the positions); every later step is
            # sync-free. check_label_capacity() - every k steps, off the hot path - raises if a batch ever exceeds it.
            frac = max(float((lm_flat != -1).sum()) / max(lm_flat.numel(), 1), float(labelled.sum()) / max(labelled.numel(), 1))
            self._auto_capacity = min(1.0, 1.2 * frac + 0.01)
            static = False
        if static:
            # Sync-free variant (HIP-graph capture, small per-GPU batches): the labelled rows are gathered into
            # FIXED-capacity buffers (torch.nonzero_static), the host never learns the counts. Padding rows carry
            # label -1 / an all-zero target: they contribute nothing to the losses or to any gradient, and the
            # divisors are device scalars. Counts above the capacity would silently drop rows, so they are recorded
            # for check_label_capacity() (GraphedTrainStep polls it without stalling the device).
            frac = self.label_capacity if self.label_capacity is not None else 0.25
            if frac == "auto":
                frac = self._auto_capacity if self._auto_capacity is not None else 0.25
            cap_t = _capacity(lm_flat.numel(), frac)
            cap_r = _capacity(labelled.numel(), frac)
            mask_t = lm_flat != -1
            idx_t = torch.nonzero_static(mask_t, size=cap_t, fill_value=0).squeeze(1)
            n_t = mask_t.sum()
            labels_t = torch.where(torch.arange(cap_t, device=idx_t.device) < n_t, lm_flat.index_select(0, idx_t),
                                   torch.full_like(idx_t, -1))
            mask_r = labelled.reshape(-1)
            idx_r = torch.nonzero_static(mask_r, size=cap_r, fill_value=0).squeeze(1)
            n_r = mask_r.sum()
            valid_r = torch.arange(cap_r, device=idx_r.device) < n_r
            self._label_counts = (n_t, n_r, cap_t, cap_r)
            # rows beyond the capacity are dropped by the gather: the divisor counts the rows actually used (so the
            # region loss stays the mean over them instead of being scaled down), and a device-side sticky flag records
            # the overflow (GraphedTrainStep reads it after every replay, check_label_capacity() on demand)
            if self._label_overflow is None or self._label_overflow.device != n_t.device:
                self._label_overflow = torch.zeros(1, dtype=torch.int32, device=n_t.device)
            self._label_overflow.logical_or_(((n_t > cap_t) | (n_r > cap_r)).reshape(1))
            divisor_r = torch.clamp(n_r, max=cap_r).to(torch.float32).reshape(1)
        else:
            idx_t = torch.nonzero(lm_flat != -1).squeeze(1)
            idx_r = torch.nonzero(labelled.reshape(-1)).squeeze(1)      # index into [B, n_reg_all - 1]
            if idx_t.numel() == 0 or idx_r.numel() == 0:
                return None
            labels_t = lm_flat.index_select(0, idx_t)
            valid_r = None
            divisor_r = float(idx_r.numel())
        idx_v = idx_r + torch.div(idx_r, per, rounding_mode="floor") + 1   # same rows inside [B, n_reg_all]

        pooled_output = _dropout(_fuse_pooled(cls.fusion_method, pooled_output_t, pooled_output_v), cls.dropout)
        seq_relationship_score = F.linear(pooled_output, cls.bi_seq_relationship.weight,
                                          cls.bi_seq_relationship.bias)
        # losses: native row kernels (csrc/loss.hip) - log-sum-exp forward, (softmax - onehot) backward
        next_sentence_loss = F.cross_entropy(seq_relationship_score.view(-1, 2), next_sentence_label.reshape(-1),
                                             ignore_index=-1)

        rows_t = sequence_output_t.reshape(-1, sequence_output_t.size(-1)).index_select(0, idx_t)
        masked_lm_loss = F.cross_entropy(cls.predictions(rows_t, pad_cols=True), labels_t, ignore_index=-1)

        rows_v = sequence_output_v.reshape(-1, sequence_output_v.size(-1)).index_select(0, idx_v)
        scores_v = cls.imagePredictions(rows_v)
        target = image_target.reshape(-1, image_target.size(-1)).index_select(0, idx_r)
        if valid_r is not None:
            target = target * valid_r.unsqueeze(1).to(target.dtype)     # padding rows: all-zero target
        if self.visual_target == 1:
            masked_img_loss = torch.sum(self.vis_criterion(scores_v, target)) / max(
                torch.sum(labelled.unsqueeze(2).expand(-1, -1, image_target.size(-1))), 1)
        else:
            # divisor: the reference's max(sum(image_label == 1), 0) (:1520-1522) = the number of rows here
            masked_img_loss = F.kl_div_log_softmax(scores_v, target, divisor_r)
        return masked_lm_loss.unsqueeze(0), masked_img_loss.unsqueeze(0), next_sentence_loss.unsqueeze(0)

    def check_label_capacity(self):
        """Raises if the last fixed-capacity label gather overflowed (synchronises; call it off the hot path)."""
        if self._label_counts is None:
            return
        if self._label_overflow is not None and int(self._label_overflow.item()) != 0:
            self._label_overflow.zero_()
            raise RuntimeError("labelled rows exceeded the fixed gather capacity in an earlier step: raise "
                               "model.label_capacity (fraction of positions, <= 1.0)")
        n_t, n_r, cap_t, cap_r = self._label_counts
        n_t, n_r = int(n_t.item()), int(n_r.item())
        if n_t > cap_t or n_r > cap_r:
            raise RuntimeError("labelled rows exceed the fixed gather capacity (%d tokens / capacity %d, %d regions / "
                               "capacity %d): raise model.label_capacity (fraction of positions, <= 1.0)"
                               % (n_t, cap_t, n_r, cap_r))

    def _nce_region_loss(self, input_ids, prediction_scores_v, image_target, labelled):
        """visual_target == 2 (:1523-1575): 70 % negatives from other samples, 30 % from the same image."""
        n_across, n_inside = int(self.num_negative * 0.7), int(self.num_negative * 0.3)
        batch_size, num_regions, _ = prediction_scores_v.size()
        assert batch_size != 0
        row_across = input_ids.new(batch_size, num_regions, n_across).random_(0, batch_size - 1)
        col_across = input_ids.new(batch_size, num_regions, n_across).random_(0, num_regions)
        for i in range(batch_size - 1):
            row_across[i][row_across[i] == i] = batch_size - 1
        row_inside = input_ids.new(batch_size, num_regions, n_inside).zero_()
        col_inside = input_ids.new(batch_size, num_regions, n_inside).random_(0, num_regions - 1)
        for i in range(batch_size):
            row_inside[i] = i
        for i in range(num_regions - 1):
            col_inside[:, i, :][col_inside[:, i, :] == i] = num_regions - 1
        final_index = torch.cat((row_across * num_regions + col_across,
                                 row_inside * num_regions + col_inside), dim=2)
        predict_v = prediction_scores_v[labelled]
        negative_v = image_target.view(batch_size * num_regions, -1)[final_index[labelled]]
        sample_v = torch.cat((image_target[labelled].unsqueeze(1), negative_v), dim=1)
        score = torch.bmm(sample_v, predict_v.unsqueeze(2)).squeeze(2)
        return self.vis_criterion(score, input_ids.new(score.size(0)).zero_())


class SimpleClassifier(nn.Module):
    """Linear -> GELU -> LayerNorm -> Linear, reference vilbert.py:1711-1722 (``logit_fc.0/2/3`` names)."""

    def __init__(self, in_dim, hid_dim, out_dim, dropout):
        super().__init__()
        self.logit_fc = nn.Sequential(nn.Linear(in_dim, hid_dim), GeLU(), BertLayerNorm(hid_dim, eps=1e-12),
                                      nn.Linear(hid_dim, out_dim))

    def forward(self, hidden_states):
        fc0, _, norm, fc3 = self.logit_fc
        return F.linear(norm(F.linear(hidden_states, fc0.weight, fc0.bias, act="gelu")), fc3.weight, fc3.bias)


class GeLU(nn.Module):
    """Parameter-free placeholder at ``logit_fc.1`` (the GELU itself is the GEMM epilogue)."""

    def forward(self, x):
        raise RuntimeError("GeLU is fused into the preceding GEMM; call SimpleClassifier.forward")


class VILBertForVLTasks(BertPreTrainedModel):
    """Multi-task wrapper: every head is computed on every forward, reference vilbert.py:1600-1708."""

    def __init__(self, config, num_labels, dropout_prob=0.1, default_gpu=True):
        super(VILBertForVLTasks, self).__init__(config)
        self.num_labels = num_labels
        self.bert = BertModel(config)
        self.dropout = nn.Dropout(dropout_prob)
        self.cls = BertPreTrainingHeads(config, self.bert.embeddings.word_embeddings.weight)
        self.vil_prediction = SimpleClassifier(config.bi_hidden_size, config.bi_hidden_size * 2, 3129, 0.5)
        self.vil_prediction_gqa = SimpleClassifier(config.bi_hidden_size, config.bi_hidden_size * 2, 1533, 0.5)
        self.vil_binary_prediction = SimpleClassifier(config.bi_hidden_size * 2, config.bi_hidden_size * 2, 2, 0.5)
        self.vil_logit = nn.Linear(config.bi_hidden_size, 1)
        self.vil_tri_prediction = nn.Linear(config.bi_hidden_size, 3)
        self.vision_logit = nn.Linear(config.v_hidden_size, 1)
        self.linguisic_logit = nn.Linear(config.hidden_size, 1)
        self.fusion_method = config.fusion_method
        self.apply(self.init_weights)
        self.tie_weights()

    def tie_weights(self):
        self._tie_or_clone_weights(self.cls.predictions.decoder, self.bert.embeddings.word_embeddings)

    def forward(self, input_txt, input_imgs, image_loc, token_type_ids=None, attention_mask=None,
                image_attention_mask=None, co_attention_mask=None, task_ids=None,
                output_all_encoded_layers=False, output_all_attention_masks=False):
        from . import ops
        sequence_output_t, sequence_output_v, pooled_output_t, pooled_output_v, all_attention_mask = self.bert(
            input_txt, input_imgs, image_loc, token_type_ids, attention_mask, image_attention_mask,
            co_attention_mask, task_ids, output_all_encoded_layers=output_all_encoded_layers,
            output_all_attention_masks=output_all_attention_masks)

        linguisic_prediction, vision_prediction, vil_binary_prediction = self.cls(
            sequence_output_t, sequence_output_v, pooled_output_t, pooled_output_v)
        pooled_output = _dropout(_fuse_pooled(self.fusion_method, pooled_output_t, pooled_output_v), self.dropout)

        vil_prediction = self.vil_prediction(pooled_output)
        vil_prediction_gqa = self.vil_prediction_gqa(pooled_output)
        if pooled_output.size(0) % 2 == 0:  # otherwise the NSP score from self.cls leaks through (:1686-1689)
            vil_binary_prediction = self.vil_binary_prediction(pooled_output.view(-1, pooled_output.size(1) * 2))
        vil_logit = F.linear(pooled_output, self.vil_logit.weight, self.vil_logit.bias)
        vil_tri_prediction = F.linear(pooled_output, self.vil_tri_prediction.weight, self.vil_tri_prediction.bias)
        # like the reference this needs image_attention_mask (:1692-1694 crash on None)
        region_mask = ops.additive_mask(image_attention_mask).unsqueeze(2)
        vision_logit = F.linear(_dropout(sequence_output_v, self.dropout), self.vision_logit.weight,
                                self.vision_logit.bias, residual=region_mask)
        linguisic_logit = F.linear(_dropout(sequence_output_t, self.dropout), self.linguisic_logit.weight,
                                   self.linguisic_logit.bias)
        return (vil_prediction, vil_prediction_gqa, vil_logit, vil_binary_prediction, vil_tri_prediction,
                vision_prediction, vision_logit, linguisic_prediction, linguisic_logit, all_attention_mask)
