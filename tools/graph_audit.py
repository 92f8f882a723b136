"""Audit of a captured training step (round-5 review item 1b: the chain graph of the fp8 training mode replayed wrong losses
once the process had freed device memory before the capture - tools/dbg_graph_nan.py).

    python tools/graph_audit.py [--mode fp8|bf16|f32|fp8+bf16] [--branches chain|fork] [--prior none|f32|fp8|bf16]
                                [--poison] [--taps] [--replays 6] [--tag NAME]

What it does, in one process:
  1. optional PRIOR work (eager training steps of another model in `--prior` mode, then dropped): the caching allocator now holds
     freed blocks - the condition under which the issue appeared;
  2. builds GraphedTrainStep in `--mode` with every native entry point wrapped: each call made WHILE CAPTURING is logged with every
     pointer-sized argument (plain arguments and the fields of the argument structs);
  3. POINTER AUDIT: every logged pointer is looked up in torch.cuda.memory_snapshot() taken after the capture - it must lie in the
     graph's private pool or in a block that is still allocated; a pointer into a FREED block of the ordinary pool is memory the
     graph will read or write at replay while the allocator hands it to somebody else;
  4. --poison: every free block of the ordinary pools is overwritten with 0xFF bytes (NaN as fp32 / bf16 / e4m3) before the
     replays - a use-after-free then shows at the first replay instead of "from the third one on, depending on the addresses";
  5. --taps: the outputs of the tensor-level launchers (vilbert/ops.py) created during the capture are kept; after every replay
     their (sum, NaN count) are written to gpurun_out/graph_audit_<tag>.txt - two runs (with / without prior work) can be diffed
     to find the first launch whose result differs;
  6. replays the graph and prints the losses next to the eager step's.
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vilbert-multi-task_amd"))
sys.path.insert(0, ROOT)
import vilbert.vilbert as V  # noqa: E402
from oracle import synth  # noqa: E402
from vilbert import _native, ops  # noqa: E402
from vilbert.graphed import GraphedTrainStep  # noqa: E402
from vilbert.optim import AdamW  # noqa: E402
from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining  # noqa: E402

DEV = "cuda:0"
NAMES = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
         "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="fp8")
ap.add_argument("--branches", default="chain")
ap.add_argument("--prior", default="f32")
ap.add_argument("--poison", action="store_true")
ap.add_argument("--taps", action="store_true")
ap.add_argument("--replays", type=int, default=6)
ap.add_argument("--tag", default=None)
ap.add_argument("--config", default="bert_base_2layer_2conect.json")
ap.add_argument("--shape", default="4,12,10")
opt = ap.parse_args()
tag = opt.tag or "%s_%s_prior-%s%s%s" % (opt.mode.replace("+", "_"), opt.branches, opt.prior, "_poison" if opt.poison else "",
                                        "_taps" if opt.taps else "")
V._drop_p = lambda m: 0.0
cfg = synth.load_config(opt.config)
sd = synth.make_state_dict(cfg, "pretraining")
shape = tuple(int(v) for v in opt.shape.split(","))


def model():
    m = BertForMultiModalPreTraining(BertConfig.from_dict(cfg))
    m.load_state_dict(sd)
    return m.to(DEV).train()


def eager_steps(mode, n, shp, lr=2e-4):
    _native.set_gemm_mode(mode)
    args = [synth.make_inputs(cfg, *shp, seed=70, with_labels=True)[k].to(DEV) for k in NAMES]
    m = model()
    o = AdamW(m.parameters(), lr=lr)
    out = []
    for _ in range(n):
        o.zero_grad()
        loss = sum(l.mean() for l in m(*args))
        loss.backward()
        o.step()
        out.append(round(loss.item(), 4))
    _native.set_gemm_mode("f32")
    return out


# ---- 2. wrap the native entry points ----------------------------------------------------------------------------------
CALLS = []          # (function, [(where, pointer)]) of every call made while capturing


def _pointers_of(arg, where, out):
    if isinstance(arg, ctypes.Structure):
        for name, _t in arg._fields_:
            _pointers_of(getattr(arg, name), "%s.%s" % (where, name), out)
    elif isinstance(arg, ctypes.Array):
        for i, v in enumerate(arg):
            _pointers_of(v, "%s[%d]" % (where, i), out)
    elif isinstance(arg, ctypes.c_void_p):
        if arg.value:
            out.append((where, int(arg.value)))
    elif isinstance(arg, int) and not isinstance(arg, bool):
        if arg >= (1 << 32):            # device addresses on this platform are far above 4 GiB; sizes / seeds can be too: filtered later
            out.append((where, arg))
    elif hasattr(arg, "_obj"):           # ctypes.byref(struct)
        _pointers_of(arg._obj, where, out)


def wrap_library():
    handle = _native.lib()
    for name in list(_native.SIGNATURES):
        fn = getattr(handle, name)

        def make(fn, name):
            def call(*args):
                if torch.cuda.is_current_stream_capturing():
                    ptrs = []
                    for i, a in enumerate(args[1:], 1):       # args[0] = the stream
                        _pointers_of(a, "arg%d" % i, ptrs)
                    CALLS.append((name, ptrs))
                return fn(*args)
            return call
        try:
            setattr(handle, name, make(fn, name))
        except Exception as exc:       # pragma: no cover
            print("could not wrap", name, exc)


TAPS = []


def wrap_ops():
    names = ["linear_fwd", "linear_bwd_input", "linear_bwd_weight", "layernorm_fwd", "layernorm_bwd", "attention_fwd", "act_bwd",
             "dropout", "quantize_rows_fp8", "text_embed_ln_fwd", "image_embed_ln_fwd", "text_embed_bwd", "xent_fwd", "xent_bwd",
             "kl_fwd", "kl_bwd"]

    def flat(x, out):
        if torch.is_tensor(x):
            out.append(x)
        elif isinstance(x, (list, tuple)):
            for v in x:
                flat(v, out)
        return out
    for nm in names:
        real = getattr(ops, nm, None)
        if real is None:
            continue

        def make(real, nm):
            def call(*a, **k):
                r = real(*a, **k)
                if torch.cuda.is_current_stream_capturing():
                    TAPS.append((nm, flat(r, [])))
                return r
            return call
        setattr(ops, nm, make(real, nm))


# ---- 3. / 4. allocator snapshot helpers ----------------------------------------------------------------------------------
def block_map():
    rows = []
    for seg in torch.cuda.memory_snapshot():
        private = tuple(seg.get("segment_pool_id", (0, 0))) != (0, 0)
        addr = seg["address"]
        for b in seg["blocks"]:
            rows.append((addr, b["size"], b["state"], private, seg.get("stream", 0)))
            addr += b["size"]
    rows.sort()
    return rows


def classify(rows, ptr):
    import bisect
    i = bisect.bisect_right(rows, (ptr, float("inf"))) - 1
    if i < 0:
        return None
    a, size, state, private, stream = rows[i]
    if ptr >= a + size:
        return None
    return ("private" if private else "ordinary") + ":" + state


def poison_free_blocks():
    torch.cuda.synchronize()
    rows = [r for r in block_map() if not r[3] and r[2] == "inactive"]
    held, hit, total = [], 0, 0
    for a, size, _state, _priv, stream in sorted(rows, key=lambda r: -r[1]):
        st = torch.cuda.ExternalStream(stream, device=DEV) if stream else torch.cuda.default_stream(DEV)
        with torch.cuda.stream(st):
            t = torch.empty(size, dtype=torch.uint8, device=DEV)
            t.fill_(0xFF)
        hit += int(a <= t.data_ptr() < a + size)
        total += size
        held.append(t)
    torch.cuda.synchronize()
    print("poison: %d free blocks (%.1f MB) of the ordinary pools overwritten with 0xFF (%d re-allocated exactly in place)"
          % (len(rows), total / 1e6, hit))
    del held
    torch.cuda.synchronize()


# ---- run -------------------------------------------------------------------------------------------------------------------
if opt.prior != "none":
    print("prior: 3 eager steps in mode %s at shape (8, 36, 37):" % opt.prior, eager_steps(opt.prior, 3, (8, 36, 37)))
wrap_library()
if opt.taps:
    wrap_ops()
_native.set_gemm_mode(opt.mode)
args = [synth.make_inputs(cfg, *shape, seed=70, with_labels=True)[k].to(DEV) for k in NAMES]
m = model()
o = AdamW(m.parameters(), lr=3e-4)
step = GraphedTrainStep(m, o, args, warmup=2, branches=opt.branches)
torch.cuda.synchronize()
rows = block_map()
bad, seen, unknown = {}, 0, {}
for fn, ptrs in CALLS:
    for where, p in ptrs:
        c = classify(rows, p)
        if c is None:
            unknown.setdefault((fn, where), 0)
            unknown[(fn, where)] += 1
            continue
        seen += 1
        if c == "ordinary:inactive":
            bad.setdefault((fn, where), []).append(p)
print("pointer audit: %d native calls captured, %d device pointers classified, %d in FREED ordinary-pool blocks"
      % (len(CALLS), seen, sum(len(v) for v in bad.values())))
for (fn, where), ps in sorted(bad.items()):
    print("   FREED: %s %s x%d (e.g. 0x%x)" % (fn, where, len(ps), ps[0]))
if unknown:
    print("   (not inside any allocator segment - host pointers, seeds, sizes: %s)"
          % ", ".join(sorted({"%s %s" % k for k in unknown})[:12]))
if opt.poison:
    poison_free_blocks()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", "graph_audit_%s.txt" % tag), "w") if opt.taps else None
got = []
for r in range(opt.replays):
    got.append(round(step(*args).item(), 4))
    if log is not None:
        torch.cuda.synchronize()
        for i, (nm, ts) in enumerate(TAPS):
            for j, t in enumerate(ts):
                f = t.double() if t.is_floating_point() else t.long().double()
                log.write("replay %d tap %04d %s out%d %s sum %.9e nan %d\n"
                          % (r, i, nm, j, tuple(t.shape), float(torch.nan_to_num(f).sum()), int(torch.isnan(f).sum())))
print("graphed %s (%s) losses: %s" % (opt.mode, opt.branches, got))
step.close()
del step, m, o
want = eager_steps(opt.mode, opt.replays, shape, lr=3e-4)
print("eager %s losses:   %s" % (opt.mode, want))
ok = all(abs(a - b) <= 2e-2 * max(1.0, abs(b)) for a, b in zip(got, want))
print("RESULT %s: %s" % (tag, "replays track the eager step" if ok else "REPLAYS DIVERGE from the eager step"))
