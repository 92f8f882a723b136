"""Benchmark of the ViLBERT two-stream hot path on MI355X (contract: see the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode fwd|train] [--batch B]

One "step" = one pass of the hot path (BertModel encoder + all heads) over one synthetic batch of
36 regions x 2048 features + 36 tokens, model bert_base_6layer_6conect.json, random (seeded) weights,
inputs resident in HBM before the timed region. Prints ONE JSON line on rank 0.

N > 1: one rank per GPU over RCCL. Either launch it as the driver does (python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...) or just run `python bench.py --gpus N`: without RANK in the environment
the script re-executes itself through torch.distributed.run on 127.0.0.1. The headline value is weak scaling (fixed
per-GPU batch 256, BASELINE.json's metric); the same line also carries the reference's own data-parallel
configuration (BASELINE configs[2]: GLOBAL batch 512 divided over the ranks, reference train_concap.py:290-294) as
"global512" and - on one GPU - the north-star forward point (batch 512) as "fwd_b512". The timed regions are
bracketed by barrier + synchronize and the max over ranks is taken.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "vilbert-multi-task_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

# the step keeps up to four HIP streams busy (+ RCCL's): they must not share hardware queues (see vilbert/__init__.py);
# read by the HIP runtime at its first call, so it has to be in the environment before torch touches the device
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BF16_DTYPE = "bf16 activations / saved tensors / activation gradients in HBM, bf16 MFMA operands, fp32 accumulate; fp32 master " \
             "weights, weight gradients, LayerNorm and softmax statistics, AdamW - NOT inside the 1e-4 parity bar " \
             "(tests/test_bf16_stream_gpu.py: its measured errors)"
BF16_NOTE = "opt-in reduced-precision TRAINING mode (round 5; the reference's model.half() + FP16_Optimizer, train_concap.py:443-461): " \
            "the encoder's hidden states, saved tensors and their gradients are bfloat16 tensors (vilbert/ops16.py, " \
            "csrc/gemm_bf16.hip), fp32 master weights with bf16 shadows refreshed once per step, fp32 weight gradients in the " \
            "arena; outside the 1e-4 parity bar (loss within 2e-2, gradient error median 3.5e-2: tests/test_bf16_stream_gpu.py, " \
            "test_gemm_modes_gpu.py); bf16 dense MFMA peak 2500 TF"
CONFIG = "bert_base_6layer_6conect.json"
N_TOK, N_REG = 36, 36
PEAK_FP32_MFMA_TFLOPS = 157.3  # gfx950 v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md


def model_flops_per_sample(cfg, T, R, heads="vltasks"):
    """Algorithmic forward FLOPs per sample (2 per MAC, no padding) - BASELINE.md section 2."""
    H, I, Hv, Iv, Hb = (cfg["hidden_size"], cfg["intermediate_size"], cfg["v_hidden_size"],
                        cfg["v_intermediate_size"], cfg["bi_hidden_size"])
    Lt, Lv, Lc = cfg["num_hidden_layers"], cfg["v_num_hidden_layers"], len(cfg["v_biattention_id"])
    V = cfg["vocab_size"]
    text = Lt * (2 * T * (4 * H * H + 2 * H * I) + 4 * T * T * H)
    image = Lv * (2 * R * (4 * Hv * Hv + 2 * Hv * Iv) + 4 * R * R * Hv)
    conn = Lc * (2 * (3 * R * Hv * Hb + 3 * T * H * Hb + R * Hb * Hv + T * Hb * H + 2 * R * Hv * Iv + 2 * T * H * I)
                 + 8 * T * R * Hb)
    emb = 2 * R * (cfg["v_feature_size"] + 5) * Hv
    pool = 2 * (H + Hv) * Hb
    bert = text + image + conn + emb + pool
    pre_heads = 2 * T * (H * H + H * V) + 2 * R * (Hv * Hv + Hv * cfg["v_target_size"]) + 2 * Hb * 2
    total = bert + pre_heads
    if heads == "vltasks":
        total += 2 * (Hb * 2 * Hb + 2 * Hb * 3129) + 2 * (Hb * 2 * Hb + 2 * Hb * 1533) \
            + (2 * (2 * Hb * 2 * Hb + 2 * Hb * 2)) / 2 + 2 * Hb * 4 + 2 * R * Hv + 2 * T * H
    return bert, total


def build_model(cfg, kind, device):
    """Random-init weights of the named architecture (the classes' own init_weights, reference
    vilbert.py:1265-1277 semantics) - there are no checkpoints on the bench box."""
    from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining, VILBertForVLTasks
    torch.manual_seed(1234)
    c = BertConfig.from_dict(cfg)
    model = VILBertForVLTasks(c, num_labels=1) if kind == "vltasks" else BertForMultiModalPreTraining(c)
    return model.to(device)


def synthetic_batch(cfg, batch, n_tok, n_reg, seed, with_labels):
    """One loader-shaped batch: full-length token / region rows, box features in
    [0, 2), normalised box coordinates, and - for the pre-training step - the ConceptCap loader's label
    conventions (reference vilbert/datasets/concept_cap_dataset.py:244-282,608-670): ~15 % of tokens and
    regions labelled, -1 elsewhere, region targets are probability rows, region 0 is the global feature."""
    g = torch.Generator().manual_seed(seed)
    V, Fv = cfg["vocab_size"], cfg["v_feature_size"]
    ids = torch.randint(0, V, (batch, n_tok), generator=g)
    ids[:, 0] = 101
    loc = torch.rand(batch, n_reg, 5, generator=g)
    loc[:, 0] = torch.tensor([0.0, 0.0, 1.0, 1.0, 1.0])
    x = dict(input_ids=ids, image_feat=torch.rand(batch, n_reg, Fv, generator=g) * 2.0, image_loc=loc,
             token_type_ids=torch.zeros(batch, n_tok, dtype=torch.long),
             attention_mask=torch.ones(batch, n_tok, dtype=torch.long),
             image_attention_mask=torch.ones(batch, n_reg, dtype=torch.long),
             co_attention_mask=torch.zeros(batch, n_reg, n_tok))
    if with_labels:
        lm = torch.where(torch.rand(batch, n_tok, generator=g) < 0.15, ids, torch.full_like(ids, -1))
        lm[:, 1] = ids[:, 1]
        il = torch.where(torch.rand(batch, n_reg - 1, generator=g) < 0.15, 1, -1)
        il[:, 0] = 1
        x.update(masked_lm_labels=lm, image_label=il,
                 image_target=torch.softmax(torch.randn(batch, n_reg - 1, cfg["v_target_size"], generator=g), -1),
                 next_sentence_label=torch.randint(0, 2, (batch,), generator=g))
    return x


def cpu_baseline(cfg, mode, budget_s=25.0):
    """The oracle (CPU restatement of the reference, oracle/vilbert_oracle.py) timed on this host's cores
    on a bounded sample of the same workload. Reported next to the GPU number, never the product path.
    Batch as BASELINE.md section 3 plans: 16 for the train step, 64 for the forward; min / median / max of the runs.
    Thread count: the best of a short calibration over {16, 32, 64} (capped by the core count) - on a
    256-core host torch's default of one thread per core is two orders of magnitude slower."""
    from oracle import synth, vilbert_oracle as vo
    train = mode == "train"
    kind = "pretraining" if train else "vltasks"
    B = 16 if train else 64
    sd = synth.make_state_dict(cfg, kind)
    x = synth.make_inputs(cfg, B, N_TOK, N_REG + (1 if train else 0), ragged=False, with_labels=train)
    if train:
        args = (x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
                x["image_attention_mask"], x["masked_lm_labels"], x["image_label"], x["image_target"],
                x["next_sentence_label"])
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "cls.predictions.decoder.weight"}
        leaves["cls.predictions.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]

        def run():
            for v in leaves.values():
                v.grad = None
            sum(l.sum() for l in vo.pretraining_forward(leaves, cfg, *args)).backward()
    else:
        args = (x["input_ids"], x["image_feat"], x["image_loc"], x["token_type_ids"], x["attention_mask"],
                x["image_attention_mask"], x["co_attention_mask"])

        def run():
            with torch.no_grad():
                vo.vltasks_forward(sd, cfg, *args)

    def once():
        t0 = time.perf_counter()
        run()
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    best_threads, best = None, None
    for th in sorted({min(t, os.cpu_count()) for t in (16, 32, 64)}):
        torch.set_num_threads(th)
        once()  # warm-up at this thread count
        t = once()
        if best is None or t < best:
            best_threads, best = th, t
        if time.perf_counter() - t_start > budget_s * 0.5:
            break
    torch.set_num_threads(best_threads)
    times = [once()]
    while len(times) < 7 and time.perf_counter() - t_start < budget_s:
        times.append(once())
    times.sort()
    med = times[len(times) // 2]
    all_cores = _cpu_all_cores(mode, B) if (os.cpu_count() or 1) > best_threads else None
    return {"value": round(B / med, 2), "unit": "samples/s", "cores": best_threads, "kind": "port", "batch": B,
            "runs": len(times), "min_median_max_s": [round(times[0], 3), round(med, 3), round(times[-1], 3)],
            "why_port": "the reference source tree (/root/reference) does not exist on the GPU box; the oracle is its "
                        "line-by-line restatement, pinned against the real reference by tests/golden + "
                        "tests/test_oracle_vs_reference.py (BASELINE.md section 3 plans the reference's own code at "
                        "batch 16 - same torch CPU ops, same threads)",
            "sample": "oracle/vilbert_oracle.py %s, batch %d, median of %d runs (min %.3fs max %.3fs); torch %s "
                      "CPU fp32, %d threads used of %d host cores" %
                      ("fwd+bwd (pre-training losses, autograd)" if train else "forward (VILBertForVLTasks, all heads)",
                       B, len(times), times[0], times[-1], torch.__version__, best_threads, os.cpu_count()),
            "all_cores": all_cores}


def _cpu_all_cores(mode, batch, limit_s=40.0):
    """SURVEY.md section 8(d) asks for `torch.set_num_threads(os.cpu_count())`: the same sample once more with one
    thread per host core, reported NEXT TO the calibrated figure (on a 256-core host torch's intra-op pool is far
    slower at these matrix sizes than a few dozen threads). Runs in a child process under a time limit so that the
    default bench line stays bounded; a run that does not finish reports the limit as an upper bound on its rate."""
    import subprocess
    cores = os.cpu_count() or 1
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-sample", mode, "--cpu-threads", str(cores), "--config", CONFIG,
           "--tokens", str(N_TOK), "--regions", str(N_REG)]
    t0 = time.perf_counter()
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s).stdout
        rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
        return {"value": round(batch / rec["seconds"], 2), "unit": "samples/s", "cores": cores, "batch": batch,
                "seconds": round(rec["seconds"], 3), "runs": 1}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "samples/s", "cores": cores, "batch": batch, "runs": 0,
                "note": "one pass did not finish within %.0f s (< %.2f samples/s)" % (limit_s, batch / limit_s)}
    except Exception as e:           # never let the baseline break the bench line
        return {"value": None, "cores": cores, "note": "failed after %.1f s: %r" % (time.perf_counter() - t0, e)}


def _cpu_sample_main(mode, threads):
    """Child of _cpu_all_cores: one warm-up + one timed pass of the oracle at `threads` threads; prints {"seconds": s}."""
    from oracle import synth, vilbert_oracle as vo
    cfg = json.load(open(os.path.join(ROOT, "vilbert-multi-task_amd", "config", CONFIG)))
    cfg = dict(synth._DEFAULTS, **cfg)
    train = mode == "train"
    B = 16 if train else 64
    sd = synth.make_state_dict(cfg, "pretraining" if train else "vltasks")
    x = synth.make_inputs(cfg, B, N_TOK, N_REG + (1 if train else 0), ragged=False, with_labels=train)
    torch.set_num_threads(threads)
    if train:
        args = tuple(x[n] for n in ("input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask",
                                    "image_attention_mask", "masked_lm_labels", "image_label", "image_target",
                                    "next_sentence_label"))
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "cls.predictions.decoder.weight"}
        leaves["cls.predictions.decoder.weight"] = leaves["bert.embeddings.word_embeddings.weight"]

        def run():
            for v in leaves.values():
                v.grad = None
            sum(l.sum() for l in vo.pretraining_forward(leaves, cfg, *args)).backward()
    else:
        args = tuple(x[n] for n in ("input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask",
                                    "image_attention_mask", "co_attention_mask"))

        def run():
            with torch.no_grad():
                vo.vltasks_forward(sd, cfg, *args)
    run()
    t0 = time.perf_counter()
    run()
    print(json.dumps({"seconds": time.perf_counter() - t0}), flush=True)


def main():
    global CONFIG, N_TOK, N_REG
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["fwd", "train"], default="train")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--config", default=CONFIG, help="model config JSON under vilbert-multi-task_amd/config/ "
                    "(default = the metric's bert_base_6layer_6conect.json; e.g. bert_large_6layer_6conect.json)")
    ap.add_argument("--tokens", type=int, default=N_TOK, help="tokens per sample (metric: 36)")
    ap.add_argument("--regions", type=int, default=N_REG, help="regions per sample (metric: 36; task shapes: 101)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", choices=["fwd", "train"], default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--gemm-mode", choices=["f32", "bf16x6", "bf16x3", "bf16", "fp8", "mxfp8"], default="f32",
                    help="GEMM arithmetic: f32 = exact fp32 MFMA (default); bf16x6 = fp32 emulated with 6 bf16 "
                         "MFMA products (fp32-class); bf16x3 = 3 products")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the extra bf16x6 measurement")
    ap.add_argument("--host-inputs", action="store_true",
                    help="(train, 1 GPU) also time the step fed from HOST numpy batches through the device input "
                         "pipeline (pinned double-buffered H2D + vb_concap_finish_batch): the PCIe-inclusive rate")
    ap.add_argument("--gemm-breakdown", action="store_true", help="print the per-shape GEMM time of the profiled step "
                    "(stderr; HIP events around every launch, single stream)")
    ap.add_argument("--graph", action="store_true", help="(train) capture the whole step - forward, backward, gradient "
                    "exchange, AdamW - into ONE HIP graph (vilbert/graphed.py) and time the replays")
    ap.add_argument("--global-batch", type=int, default=0, help="GLOBAL batch divided over the ranks (strong scaling; the "
                    "reference's own data-parallel mode, train_concap.py:290-294) instead of a fixed per-GPU batch")
    ap.add_argument("--label-gather", choices=["auto", "exact"], default="auto",
                    help="(train) gather of the labelled rows in front of the pre-training heads: auto = fixed capacity fixed by "
                         "the first step's counts, no host sync per step (overflow checked after every timed region); exact = "
                         "torch.nonzero per step (one host sync)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the global512 / fwd_b512 legs of the default line")
    ap.add_argument("--deterministic", action="store_true", help="deterministic split-K weight gradients "
                    "(vb_set_deterministic: workspace + ordered reduce instead of atomics; single stream)")
    ap.add_argument("--ddp-algorithm", choices=["ring", "direct"], default=os.environ.get("VB_DDP_ALGORITHM", "ring"),
                    help="gradient exchange per bucket at N > 1: ring = one all_reduce, direct = reduce_scatter + "
                         "all_gather over all xGMI links (vilbert/distributed.py); the N > 1 line times BOTH, this picks "
                         "the one the headline uses")
    ap.add_argument("--force-ddp", action="store_true", help="wrap in DistributedDataParallel even at world size 1 "
                    "(exercises the RCCL bucket path on a single GPU)")
    args = ap.parse_args()
    if args.cpu_sample:
        CONFIG, N_TOK, N_REG = args.config, args.tokens, args.regions
        return _cpu_sample_main(args.cpu_sample, args.cpu_threads or (os.cpu_count() or 1))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # not under a launcher: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1)
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: only %d GPU(s) visible" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)

    from vilbert import _native, ops
    from vilbert.vilbert import BertConfig
    _native.set_gemm_mode(args.gemm_mode)
    if args.deterministic:
        _native.set_deterministic(True, device=device)
    CONFIG, N_TOK, N_REG = args.config, args.tokens, args.regions
    cfg = BertConfig.from_json_file(os.path.join(ROOT, "vilbert-multi-task_amd", "config", CONFIG)).to_dict()
    B = args.batch
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be divisible by the number of ranks")
        B = args.global_batch // world

    def forward_workload(batch, graph=False):
        """(step, inputs dict, model): VILBertForVLTasks forward, eval + no_grad (graph: replayed as one HIP graph)."""
        xb = synthetic_batch(cfg, batch, N_TOK, N_REG, 7 + rank, False)
        names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask",
                 "image_attention_mask", "co_attention_mask"]
        inp = tuple(xb[n].to(device) for n in names)
        net = build_model(cfg, "vltasks", device).eval()

        if graph:
            from vilbert.graphed import GraphedForward
            gf = GraphedForward(net, inp)
            return (lambda: gf(*inp)), xb, net

        def fstep():
            with torch.no_grad():
                return net(*inp)
        return fstep, xb, net

    train_state = {}

    def train_workload(batch, graph=False):
        """train_concap.py step (reference :523-585): BertForMultiModalPreTraining in train mode (dropout on), the
        loader's shapes (36 regions + 1 global-mean region row -> R = 37, 36 tokens, 36x1601 region targets), loss =
        masked-LM + masked-region KL + alignment, backward, gradient all-reduce (N > 1), AdamW step. The model /
        optimizer are built once and shared by the legs that only change the batch."""
        xb = synthetic_batch(cfg, batch, N_TOK, N_REG + 1, 7 + rank, True)
        names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask",
                 "image_attention_mask", "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
        inp = tuple(xb[n].to(device) for n in names)
        if not train_state:
            net = build_model(cfg, "pretraining", device).train()
            if world > 1 or args.force_ddp:
                from vilbert.distributed import DistributedDataParallel
                net = DistributedDataParallel(net, algorithm=args.ddp_algorithm,
                                              direct_at_world_size_one=args.force_ddp and args.ddp_algorithm == "direct")
            decay = [p for n, p in net.named_parameters() if p.requires_grad and not any(
                k in n for k in ("bias", "LayerNorm.bias", "LayerNorm.weight"))]
            no_decay = [p for n, p in net.named_parameters() if p.requires_grad and any(
                k in n for k in ("bias", "LayerNorm.bias", "LayerNorm.weight"))]
            from vilbert.optim import AdamW   # native multi-tensor launch, pytorch-transformers 1.0.0 semantics
            base_net = net.module if hasattr(net, "module") else net
            if args.label_gather == "auto":
                base_net.label_capacity = "auto"
            train_state["model"] = net
            train_state["opt"] = AdamW([{"params": decay, "weight_decay": 0.01}, {"params": no_decay, "weight_decay": 0.0}],
                                       lr=1e-4, betas=(0.9, 0.98))   # train_concap.py:465-470
        net, optim = train_state["model"], train_state["opt"]

        def tstep():
            optim.zero_grad(set_to_none=True)
            lm, img, nsp = net(*inp)
            loss = lm.mean() + img.mean() + nsp.mean()
            loss.backward()
            optim.step()
            return loss
        if graph:
            # one hipGraphLaunch per step: fixed-capacity label gather, device-side dropout counter, static tables
            from vilbert.graphed import GraphedTrainStep
            gs = GraphedTrainStep(net, optim, inp)
            train_state.setdefault("graphs", []).append(gs)
            return (lambda: gs(*inp)), xb, net
        return tstep, xb, net

    def gemm_family_tf(fn):
        """TFLOP/s of all GEMM launches of one extra single-stream call of fn (HIP events around every launch)."""
        from vilbert import autograd_ops as _ao2
        from vilbert import layers as _ly2
        from vilbert import vilbert as _vb2
        two_, ws_ = _vb2.set_two_streams(False), _ao2.set_wgrad_stream(False)
        nat_ = _ly2.set_native(False)      # per-launch events live in the per-op launchers (same kernels, same order)
        try:
            fn()
            torch.cuda.synchronize()
            ops.profile_linear(True)
            fn()
            torch.cuda.synchronize()
            ms, fl, n = ops.profile_linear(False)
        finally:
            _vb2.set_two_streams(two_)
            _ao2.set_wgrad_stream(ws_)
            _ly2.set_native(nat_)
        return (fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0), n

    def bf16_roofline(tf, n):
        return {"bound": "mfma", "kernel": "gemm_bf16_kernel (forward + dgrad through the transposed weight shadow) + "
                "wgrad_bf16_kernel (ds_read_b64_tr_b16 fragments), v_mfma_f32_32x32x16_bf16, persistent 256x128 tiles: 8 MFMA + 4 "
                "LDS-DMA loader waves per CU; the heads / poolers on gemm_planes_kernel (fp32 tensors, bf16 operands)",
                "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
                "launches_per_step": n, "traffic": bf16_traffic()[0], "traffic_note": bf16_traffic()[1],
                "what": "all GEMM launches of one extra step (fwd + dgrad + wgrad), algorithmic 2MNK FLOPs / sum of HIP-event "
                        "durations (single stream); bf16 dense MFMA peak 2500 TF"}

    def bf16_traffic():
        """HBM-side bytes of the dominant bf16 symbol - the weight gradient, wgrad_bf16_kernel + its ordered reduce - at its
        most frequent large shape, from the committed rocprofv3 PMC passes (profiles/r06_bf16_gemm_traffic.json)."""
        tp = os.path.join(ROOT, "profiles", "r06_bf16_gemm_traffic.json")
        if not os.path.isfile(tp):
            return None, "not measured"
        rows = [r for r in json.load(open(tp))["launches"] if r["kernel"] == "wgrad_bf16_kernel" and r["shape"] == "text FFN up"]
        if not rows:
            return None, "not measured"
        r = rows[0]
        tot = r.get("pair_hbm_bytes_corrected", r["hbm_bytes_corrected"])
        return tot, ("rocprofv3 PMC, wgrad_bf16_kernel + wgrad_bf16_reduce_kernel (deterministic weight gradient) M=%d N=%d K=%d: "
                     "%.0f MB per launch pair vs %.0f MB algorithmic; the q|k|v forward of gemm_bf16_kernel: 147 MB vs 60 MB "
                     "(profiles/r06_gemm_pmc.txt)" % (r["M"], r["N"], r["K"], tot / 1e6, r["algorithmic_bytes"] / 1e6))

    def comm_model(step64_ms, mibs=(64, 256)):
        """(In the bf16 GEMM mode - switched on by the caller - the wrapper defaults to bf16 exchange buckets and the direct
        algorithm: half the bytes on the links.) What the gradient exchange will cost at N = 8, from what CAN be measured on one GPU: the real bucket layout
        (vilbert/distributed.py, buckets of 64 MiB = the default since round 3 and of 256 MiB over the gradient arena), the moment each bucket's all-reduce is
        launched inside backward (HIP events, world-size-1 RCCL group) and therefore the window of backward work left to
        hide it, next to SURVEY.md section 8(e)'s xGMI cost model (7 links x ~153 GB/s per GPU: ring all-reduce
        2 (N-1)/N S / 153 GB/s, direct reduce-scatter + all-gather over all links 2 (S/N) / 153 GB/s). A MODEL, recorded
        so that the prediction is on file when an 8-GPU node measures it."""
        from vilbert.distributed import DistributedDataParallel
        from vilbert.optim import AdamW
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29547")
            dist.init_process_group(backend="nccl", device_id=device, rank=0, world_size=1)
        names = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask",
                 "image_attention_mask", "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
        link = 153e9
        out = {"what": "model (not a measurement of N > 1): bucket sizes and launch times measured on one GPU, xGMI cost from "
                       "SURVEY.md 8(e)", "n_gpus_modelled": 8}
        for mib in mibs:        # 64 MiB = the wrapper's default since round 3, 256 MiB = rounds 1-2
            net = DistributedDataParallel(build_model(cfg, "pretraining", device).train(), message_size=mib * (1 << 20) // 4)
            out["bucket_dtype"] = str(net.bucket_dtype) if net.bucket_dtype is not None else "torch.float32"
            out["algorithm_default"] = net.algorithm
            optim = AdamW(net.parameters(), lr=1e-4, betas=(0.9, 0.98))
            res = {"n_buckets": len(net._buckets)}
            for pb in (64, 256):
                xb = synthetic_batch(cfg, pb, N_TOK, N_REG + 1, 11, True)
                inp = tuple(xb[n].to(device) for n in names)

                def one(trace):
                    optim.zero_grad(set_to_none=True)
                    lm, img, nsp = net(*inp)
                    loss = lm.mean() + img.mean() + nsp.mean()
                    net.trace = [] if trace else None
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    loss.backward()
                    e1.record()
                    optim.step()
                    return e0, e1
                for _ in range(3):
                    one(False)
                e0, e1 = one(True)
                torch.cuda.synchronize()
                bwd_ms = e0.elapsed_time(e1)
                rows, t_ring, t_direct = [], 0.0, 0.0
                for idx, nbytes, ev in net.trace:
                    if net.bucket_dtype is torch.bfloat16:
                        nbytes //= 2                                     # (the trace reports the fp32 arena range)
                    ready = e0.elapsed_time(ev)
                    ring = 2.0 * 7 / 8 * nbytes / link * 1e3
                    direct = 2.0 * (nbytes / 8) / link * 1e3
                    t_ring = max(t_ring, ready) + ring
                    t_direct = max(t_direct, ready) + direct
                    rows.append({"bucket": idx, "mbytes": round(nbytes / 1e6, 1), "launched_ms_into_backward": round(ready, 2),
                                 "backward_left_ms": round(bwd_ms - ready, 2), "ring_ms_n8": round(ring, 2),
                                 "direct_ms_n8": round(direct, 2)})
                net.trace = None
                res["per_gpu_batch_%d" % pb] = {
                    "backward_ms": round(bwd_ms, 2), "exposed_ms_ring": round(max(0.0, t_ring - bwd_ms), 2),
                    "exposed_ms_direct": round(max(0.0, t_direct - bwd_ms), 2),
                    "buckets": rows if mib == 256 or pb == 64 else "(%d buckets, same layout as batch 64)" % len(rows)}
            e = res["per_gpu_batch_64"]
            res["predicted_global512_n8"] = {
                "step_ms_one_gpu_b64": round(step64_ms, 2),
                "samples_per_s_ring": round(512 / ((step64_ms + e["exposed_ms_ring"]) * 1e-3), 1),
                "samples_per_s_direct": round(512 / ((step64_ms + e["exposed_ms_direct"]) * 1e-3), 1)}
            out["buckets_%d_mib" % mib] = res
            net.arena.release()
            del net, optim
            torch.cuda.empty_cache()
        out["note"] = ("8 x 64 samples per step; the all-reduces run one after the other on RCCL's stream from the moment each "
                       "bucket is launched, what is not finished when backward ends is exposed; ring = per-link bound "
                       "2 (N-1)/N S / 153 GB/s, direct = reduce-scatter + all-gather over all 7 links 2 (S/N) / 153 GB/s")
        return out


    def large_legs():
        """BASELINE configs[3]: bert_large_6layer_6conect.json (24 text layers, H = 1024, I = 4096, 16 x 64 heads) - forward
        at the metric's shape and at a real task shape (T = 24, R = 101), and the train_concap step; each with the TFLOP/s
        of its GEMM family (HIP events around every launch of one extra single-stream call)."""
        from vilbert.optim import AdamW
        lcfg = BertConfig.from_json_file(os.path.join(ROOT, "vilbert-multi-task_amd", "config",
                                                      "bert_large_6layer_6conect.json")).to_dict()
        out = {}
        with torch.device(device):
            net = build_model(lcfg, "vltasks", device).eval()
        fnames = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
                  "co_attention_mask"]
        for tag, (pb, T, R) in (("large_fwd_b256", (256, 36, 36)), ("large_fwd_tasks_b128", (128, 24, 101))):
            xb = synthetic_batch(lcfg, pb, T, R, 7, False)
            inp = tuple(xb[n].to(device) for n in fnames)

            def f():
                with torch.no_grad():
                    return net(*inp)
            n = max(4, args.steps // 2)
            dt = timed(f, 2, n)
            tf, _ = gemm_family_tf(f)
            _, tot = model_flops_per_sample(lcfg, T, R, "vltasks")
            mtf = pb * n / dt * tot / 1e12
            out[tag] = {"value": round(pb * n / dt, 2), "unit": "samples/s", "ms_per_step": round(1e3 * dt / n, 3), "steps": n,
                        "batch": pb, "tokens": T, "regions": R, "gemm_family_tflops": round(tf, 1),
                        "gemm_frac_of_fp32_mfma_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "model_tflops": round(mtf, 1),
                        "note": "bert_large_6layer_6conect.json forward (VILBertForVLTasks, eval, all heads), one GPU"}
            # the same forward in the MX inference mode (BASELINE configs[4] on configs[3]'s model); at the task shape the 101-key
            # attention rows run on the key-tiled bf16 kernel, the 24-token text rows on the MX attention kernel
            _native.set_gemm_mode("mxfp8")
            try:
                dt8 = timed(f, 2, n)
            finally:
                _native.set_gemm_mode("f32")
            out[tag + "_mxfp8"] = {"value": round(pb * n / dt8, 2), "unit": "samples/s", "ms_per_step": round(1e3 * dt8 / n, 3),
                                   "steps": n, "batch": pb, "tokens": T, "regions": R, "speedup_vs_fp32": round(dt / dt8, 2),
                                   "model_tflops": round(pb * n / dt8 * tot / 1e12, 1), "dtype": "mxfp8 linears, bf16 attention",
                                   "note": "the same forward with set_gemm_mode('mxfp8')"}
        del net
        torch.cuda.empty_cache()
        with torch.device(device):
            net = build_model(lcfg, "pretraining", device).train()
        decay = [p for n_, p in net.named_parameters() if not any(k in n_ for k in ("bias", "LayerNorm.bias", "LayerNorm.weight"))]
        no_decay = [p for n_, p in net.named_parameters() if any(k in n_ for k in ("bias", "LayerNorm.bias", "LayerNorm.weight"))]
        optim = AdamW([{"params": decay, "weight_decay": 0.01}, {"params": no_decay, "weight_decay": 0.0}], lr=1e-4, betas=(0.9, 0.98))
        tnames = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask",
                  "masked_lm_labels", "image_label", "image_target", "next_sentence_label"]
        xb = synthetic_batch(lcfg, 256, N_TOK, N_REG + 1, 7, True)
        inp = tuple(xb[n].to(device) for n in tnames)

        def t():
            optim.zero_grad(set_to_none=True)
            lm, img, nsp = net(*inp)
            (lm.mean() + img.mean() + nsp.mean()).backward()
            optim.step()
        n = max(3, args.steps // 3)
        dt = timed(t, 2, n)
        tf, launches = gemm_family_tf(t)
        out["large_train_b256"] = {"value": round(256 * n / dt, 2), "unit": "samples/s", "ms_per_step": round(1e3 * dt / n, 3),
                                   "steps": n, "batch": 256, "tokens": N_TOK, "regions": N_REG + 1,
                                   "gemm_family_tflops": round(tf, 1), "gemm_launches_per_step": launches,
                                   "gemm_frac_of_fp32_mfma_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                                   "note": "bert_large_6layer_6conect.json train_concap step (fwd + bwd, dropout on, native "
                                           "losses, AdamW), one GPU"}
        if optim._arena is not None:
            optim._arena.release()
        del net, optim
        return out

    def task_legs():
        """Fine-tuning steps at the reference's task shapes (BASELINE configs[3] direction, on the base 6L/6C model): the
        VILBertForVLTasks forward + backward + AdamW with the losses of vilbert/task_utils.py:325-341 -
          * vqa: batch 128, 23 tokens, 101 regions (vilbert_tasks.yml:12-14), BCEWithLogits on the 3,129-way answer head,
            loss.mean() * target.size(1);
          * retrieval: batch 64 x 4 candidate captions = 256 sequences, 30 tokens, 101 regions (vilbert_tasks.yml TASK7/8),
            CrossEntropy over the 4 vil_logit scores of a group -
        once on the fp32 kernels and once in the bf16 training mode (the key-ragged attention launches: 101 keys per row)."""
        from vilbert.optim import AdamW
        out = {}
        with torch.device(device):
            net = build_model(cfg, "vltasks", device).train()
        decay = [p for n_, p in net.named_parameters() if not any(k in n_ for k in ("bias", "LayerNorm.bias", "LayerNorm.weight"))]
        no_decay = [p for n_, p in net.named_parameters() if any(k in n_ for k in ("bias", "LayerNorm.bias", "LayerNorm.weight"))]
        optim = AdamW([{"params": decay, "weight_decay": 0.01}, {"params": no_decay, "weight_decay": 0.0}], lr=4e-5)
        fnames = ["input_ids", "image_feat", "image_loc", "token_type_ids", "attention_mask", "image_attention_mask"]
        g = torch.Generator().manual_seed(11)
        bce = torch.nn.BCEWithLogitsLoss(reduction="mean")
        xe = torch.nn.CrossEntropyLoss()
        for tag, (pb, T, R) in (("vqa_b128", (128, 23, 101)), ("retrieval_b64x4", (256, 30, 101))):
            xb = synthetic_batch(cfg, pb, T, R, 7, False)
            inp = tuple(xb[n].to(device) for n in fnames)
            if tag.startswith("vqa"):
                target = torch.zeros(pb, 3129)
                target[torch.arange(pb), torch.randint(0, 3129, (pb,), generator=g)] = 1.0
                target = target.to(device)
            else:
                target = torch.zeros(pb // 4, dtype=torch.long, device=device)

            def t():
                optim.zero_grad(set_to_none=True)
                o = net(*inp)
                if tag.startswith("vqa"):
                    loss = bce(o[0], target).mean() * target.size(1)
                else:
                    loss = xe(o[2].view(pb // 4, 4), target)
                loss.backward()
                optim.step()
                return loss
            for mode in ("f32", "bf16"):
                _native.set_gemm_mode(mode)
                try:
                    n = max(6, args.steps // 2)
                    dt = timed(t, 3, n)
                    tf, launches = gemm_family_tf(t)
                finally:
                    _native.set_gemm_mode("f32")
                peak = PEAK_FP32_MFMA_TFLOPS if mode == "f32" else 2500.0
                out["task_train_%s_%s" % (tag, mode)] = {
                    "value": round(pb * n / dt, 2), "unit": "sequences/s", "ms_per_step": round(1e3 * dt / n, 3), "steps": n,
                    "batch": pb, "tokens": T, "regions": R, "gemm_family_tflops": round(tf, 1), "gemm_launches_per_step": launches,
                    "gemm_frac_of_mfma_peak": round(tf / peak, 4), "dtype": "f32" if mode == "f32" else BF16_DTYPE,
                    "note": "VILBertForVLTasks fine-tuning step (fwd + bwd + AdamW, dropout on) at the %s task shape, one GPU" %
                            ("VQA (BCEWithLogits, 3,129 answers)" if tag.startswith("vqa") else
                             "image-retrieval (4 captions per image, CrossEntropy over vil_logit)")}
        if optim._arena is not None:
            optim._arena.release()
        del net, optim
        return out

    if args.mode == "fwd":
        step, x, model = forward_workload(B)
        n_reg = N_REG
    else:
        step, x, model = train_workload(B, graph=args.graph)
        n_reg = N_REG + 1
        opt = train_state["opt"]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, warmup, steps):
        """warmup untimed calls, then `steps` calls bracketed by barrier + synchronize; max over the ranks (seconds)."""
        for _ in range(warmup):
            fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # fixed-capacity label gather (--label-gather auto): a batch that exceeded the capacity would have dropped rows -
        # checked here, outside the timed region (raises)
        m_ = train_state.get("model")
        if m_ is not None:
            m_ = m_.module if hasattr(m_, "module") else m_
            if getattr(m_, "label_capacity", None) is not None:
                m_.check_label_capacity()
        return dt

    elapsed = timed(step, args.warmup, args.steps)
    label_gather_info = None
    if args.mode == "train":
        bm = train_state.get("model")
        bm = bm.module if hasattr(bm, "module") else bm
        cap = getattr(bm, "_auto_capacity", None)
        label_gather_info = "exact (torch.nonzero, one host sync per step)" if args.label_gather == "exact" or cap is None else \
            "fixed capacity %.3f of the positions (1.2 x the first step's labelled fraction + 0.01), no host sync per step; " \
            "overflow checked after every timed region" % cap

    # Roofline of the dominant kernel (the fp32-MFMA GEMM family, ~99 % of the FLOPs): every
    # vb_linear_fwd launch of extra profiled steps is bracketed with HIP events on the launch stream.
    prof_steps = 1 if args.mode == "train" else 2
    # The profiled step runs on ONE stream: with the text / image streams (and the weight-gradient side streams)
    # overlapped an event pair would time a GEMM that shares the chip with other streams' kernels, not the kernel itself.
    from vilbert import autograd_ops as _ao
    from vilbert import vilbert as _vb
    two = _vb.set_two_streams(False)
    ws_prev = _ao.set_wgrad_stream(False)
    from vilbert import layers as _ly
    nat_prev = _ly.set_native(False)       # per-launch events live in the per-op launchers (same kernels, same order)
    pstep = step
    if args.graph and args.mode == "train":
        pstep = train_workload(B)[0]      # per-launch events need eager launches (same model, same optimizer)
    pstep()
    torch.cuda.synchronize()
    ops.profile_linear(True)
    for _ in range(prof_steps):
        pstep()
    torch.cuda.synchronize()
    gemm_ms, gemm_flops, gemm_launches = ops.profile_linear(False)
    _vb.set_two_streams(two)
    _ao.set_wgrad_stream(ws_prev)
    _ly.set_native(nat_prev)
    if args.gemm_breakdown and rank == 0:
        for tag, n, ms, tf in ops.profile_breakdown():
            print("gemm %-6s M=%6d N=%6d K=%6d nseg=%d  x%3d  %8.3f ms  %6.1f TF" % (tag + (n, ms, tf)), file=sys.stderr)

    # The same workload with the opt-in bf16x6 GEMM mode (fp32 operands split into 3 bf16 planes, six MFMA
    # products per fp32 product; passes the same parity tests) - reported beside the primary number.
    alt = None
    if args.gemm_mode == "f32" and world == 1 and not args.no_alt_mode and not args.graph:
        alt = {}
        notes = {"bf16x6": "opt-in (--gemm-mode bf16x6 / VB_GEMM_MODE): fp32 operands split into 3 bf16 planes, 6 MFMA "
                           "products, fp32-class result - the same parity tests pass; GEMM ceiling 2500/6 = 417 TF",
                 "bf16": BF16_NOTE}
        notes["fp8"] = "opt-in: FORWARD linears on OCP e4m3 operands (csrc/fp8.hip), backward GEMMs exact fp32 on the saved " \
                       "fp32 activations (straight-through); outside the 1e-4 parity bar (tests/test_fp8_gpu.py)"
        notes["fp8+bf16"] = "opt-in: fp8 forward linears (as 'fp8') + bf16-operand backward GEMMs (as 'bf16'), fp32 master " \
                            "weights, accumulation, LayerNorm, attention and optimizer - throughput data point, no " \
                            "convergence claim; measured gradient error vs exact fp32: tests/test_gemm_modes_gpu.py"
        # (the "fp8" training mode - fp8 forward + exact-fp32 backward - is no longer a leg: BENCH_r02 showed it at the
        #  rate of bf16x6; `--gemm-mode fp8` still runs it. Gradient errors of the two reduced-precision modes below:
        #  tests/test_gemm_modes_gpu.py::test_reduced_precision_training_modes_report_their_gradient_error)
        for mode in ("bf16x6", "bf16", "fp8+bf16"):
            _native.set_gemm_mode(mode)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            alt_steps = max(10, args.steps) if mode == "bf16" else max(3, args.steps // 2)
            for _ in range(alt_steps):
                step()
            torch.cuda.synchronize()
            alt_ms = 1e3 * (time.perf_counter() - ta) / alt_steps
            alt[mode] = {"value": round(B / (alt_ms * 1e-3), 2), "unit": "samples/s", "ms_per_step": round(alt_ms, 3),
                         "steps": alt_steps, "note": notes[mode]}
            if mode == "bf16":
                # its own roofline object: every GEMM launch of one more (single-stream) step under HIP events, against the
                # dense bf16 MFMA peak
                tf16, n16 = gemm_family_tf(step)
                alt[mode]["dtype"] = BF16_DTYPE
                alt[mode]["roofline"] = bf16_roofline(tf16, n16)
                b64_16, _, _ = train_workload(64)
                d64 = timed(b64_16, 3, 10)
                alt[mode]["b64"] = {"value": round(64 * 10 / d64, 2), "unit": "samples/s", "ms_per_step": round(1e3 * d64 / 10, 3),
                                    "steps": 10, "eager": round(64 * 10 / d64, 2),
                                    "note": "the same mode at the reference's per-GPU batch 64 (global 512 over 8 GPUs): eager "
                                            "launches (host-bound: ~20 - 24 ms of enqueue per step)"}
                del b64_16
        _native.set_gemm_mode("f32")

    # PCIe-inclusive leg (opt-in): raw worker-shaped numpy batches on the host -> pinned staging -> async H2D on
    # a copy stream -> native batch finishing -> the same step. Reported beside `value`, never as `value`.
    host_leg = None
    if args.host_inputs and args.mode == "train" and world == 1:
        import numpy as np
        from vilbert.input_pipeline import DeviceBatchPipeline
        g = np.random.RandomState(5)
        raws = []
        for _ in range(2):
            ids = g.randint(0, cfg["vocab_size"], size=(B, N_TOK)).astype(np.int64)
            lm = np.where(g.rand(B, N_TOK) < 0.15, ids, -1).astype(np.int64)
            lm[:, 1] = ids[:, 1]
            il = np.where(g.rand(B, N_REG) < 0.15, 1, -1).astype(np.int64)
            il[:, 0] = 1
            tgt = g.rand(B, N_REG, cfg["v_target_size"]).astype(np.float32)
            tgt /= tgt.sum(-1, keepdims=True)
            raws.append((ids, np.ones((B, N_TOK), np.int64), np.zeros((B, N_TOK), np.int64), lm,
                         np.zeros(B, np.int64), g.rand(B, N_REG, cfg["v_feature_size"]).astype(np.float32) * 2,
                         g.rand(B, N_REG, 5).astype(np.float32), tgt, il, np.ones((B, N_REG), np.int64),
                         (il == 1).astype(np.int64)))
        n_host = max(4, args.steps // 2)

        def run(n):
            for b in DeviceBatchPipeline((raws[i % 2] for i in range(n)), device, objective=1):
                input_ids, input_mask, segment_ids, lm_label_ids, is_next, feat, loc, target, ilabel, imask = b
                opt.zero_grad(set_to_none=True)
                lm_l, img_l, nsp_l = model(input_ids, feat, loc, segment_ids, input_mask, imask, lm_label_ids, ilabel,
                                           target, is_next)      # argument order of train_concap.py:542-553
                (lm_l.mean() + img_l.mean() + nsp_l.mean()).backward()
                opt.step()
        run(2)
        torch.cuda.synchronize()
        th = time.perf_counter()
        run(n_host)
        torch.cuda.synchronize()
        host_ms = 1e3 * (time.perf_counter() - th) / n_host
        mb = sum(a.nbytes for a in raws[0]) / 1e6
        host_leg = {"value": round(B / (host_ms * 1e-3), 2), "unit": "samples/s", "ms_per_step": round(host_ms, 3),
                    "steps": n_host, "host_mb_per_step": round(mb, 1),
                    "note": "inputs start as numpy arrays on the host each step (pinned double-buffered H2D on a copy "
                            "stream + vb_concap_finish_batch); PCIe-inclusive, not the headline"}

    # Extra legs of the default line (same process, after the headline timing):
    #  * global512 - BASELINE configs[2] as the reference runs it: GLOBAL batch 512 split over the ranks (64 per GPU at
    #    N = 8, reference train_concap.py:290-294) - strong scaling, reported beside the weak-scaling headline;
    #  * fwd_b512 (one GPU) - the north-star target point: 6L/6C co-attention forward at batch 512.
    extra = {}
    default_line = args.mode == "train" and CONFIG == "bert_base_6layer_6conect.json" and not args.global_batch \
        and args.batch == 256 and args.gemm_mode == "f32" and not args.no_extra_legs
    if world > 1 and args.mode == "train" and not args.graph:
        # the other exchange algorithm on the same model / buckets / batch (the attribute is read per bucket launch)
        ddp_net = train_state["model"]
        other = "direct" if ddp_net.algorithm == "ring" else "ring"
        n_o = max(3, args.steps // 2)
        ddp_net.algorithm = other
        try:
            o_dt = timed(step, 2, n_o)
        finally:
            ddp_net.algorithm = args.ddp_algorithm
        extra["ddp_algorithms"] = {
            args.ddp_algorithm: {"value": round(B * world * args.steps / elapsed, 2), "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                                 "steps": args.steps, "headline": True},
            other: {"value": round(B * world * n_o / o_dt, 2), "ms_per_step": round(1e3 * o_dt / n_o, 3), "steps": n_o},
            "unit": "samples/s", "buckets": len(ddp_net._buckets),
            "note": "same step, gradient buckets exchanged with ring = one all_reduce per bucket vs direct = "
                    "reduce_scatter_tensor + all_gather_into_tensor in place on the arena range (vilbert/distributed.py)"}
    if default_line and 512 % world == 0:
        gstep, gx, _ = train_workload(512 // world)
        n_g = max(10, args.steps)
        g_dt = timed(gstep, 2, n_g)
        extra["global512"] = {"value": round(512 * n_g / g_dt, 2), "unit": "samples/s", "global_batch": 512,
                              "per_gpu_batch": 512 // world, "ms_per_step": round(1e3 * g_dt / n_g, 3), "steps": n_g,
                              "scaling": "strong", "note": "BASELINE configs[2]: train_concap step at GLOBAL batch 512 "
                              "divided over the ranks (reference train_concap.py:290-294)"}
        del gstep, gx
    if default_line and world == 1:
        # the reference's per-GPU batch (512 / 8 GPUs = 64), eager launches. (The whole-step HIP graph of
        # vilbert/graphed.py is NOT part of the default line any more: BENCH_r02 measured it 3-6 % slower than eager at
        # this batch on this host - the step is GPU-bound, ~34 ms of kernels - so it stays an option for slower hosts:
        # `bench.py --graph --batch 64`.)
        e64, _, _ = train_workload(64)
        n64 = max(6, args.steps)
        e_dt = timed(e64, 5, n64)
        extra["b64"] = {"value": round(64 * n64 / e_dt, 2), "unit": "samples/s", "per_gpu_batch": 64, "steps": n64,
                        "ms_per_step": round(1e3 * e_dt / n64, 3),
                        "note": "train_concap step at the reference's per-GPU batch 64 (BASELINE configs[2] per GPU), eager"}
        del e64
        extra["comm_model"] = comm_model(1e3 * e_dt / n64)
        if alt is not None and "bf16" in alt:
            # the bf16 mode at batch 64 with the whole step - shadow refresh, forward, backward, AdamW - replayed as ONE HIP graph
            # (vilbert/graphed.py GraphedTrainStep, captured as a chain): the eager step is bound by the host's enqueue rate.
            # Measured last of the train legs: a capture leaves its private memory pool behind.
            _native.set_gemm_mode("bf16")
            try:
                g64_16, _, _ = train_workload(64, graph=True)
                dg64 = timed(g64_16, 3, 10)
            finally:
                _native.set_gemm_mode("f32")
            b64_ = alt["bf16"]["b64"]
            b64_["graphed"] = round(64 * 10 / dg64, 2)
            if b64_["graphed"] > b64_["value"]:
                b64_["value"], b64_["ms_per_step"] = b64_["graphed"], round(1e3 * dg64 / 10, 3)
            b64_["note"] += " vs the whole step replayed as one HIP graph (GraphedTrainStep, chain form); value = the faster"
            for gs_ in train_state.pop("graphs", []):
                gs_.close()
            del g64_16
            # the exchange of the bf16 step: bf16 buckets + the direct algorithm (the wrapper's defaults in this mode)
            _native.set_gemm_mode("bf16")
            try:
                b64_["comm_model"] = comm_model(b64_["ms_per_step"], mibs=(64,))
            finally:
                _native.set_gemm_mode("f32")
        # the train model, its optimizer state and arena are not needed below
        for k in list(train_state):
            train_state.pop(k)
        torch.cuda.empty_cache()
        fstep, _, fmodel = forward_workload(512)
        n_f = max(5, args.steps // 2)
        f_dt = timed(fstep, 2, n_f)
        _, f_total = model_flops_per_sample(cfg, N_TOK, N_REG, "vltasks")
        f_tf = 512 * n_f / f_dt * f_total / 1e12
        extra["fwd_b512"] = {"value": round(512 * n_f / f_dt, 2), "unit": "samples/s", "ms_per_step": round(1e3 * f_dt / n_f, 3),
                             "steps": n_f, "model_tflops": round(f_tf, 2),
                             "frac_of_fp32_mfma_peak": round(f_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                             "note": "north-star target point: VILBertForVLTasks forward (eval, no_grad, all heads), "
                                     "batch 512, T = R = 36, one GPU; target >= 0.40 of the MFMA peak"}
        if not args.no_cpu_baseline:
            # north_star: "next to the reference CPU forward timed on the node's own host cores (core count stated)"
            extra["fwd_b512"]["cpu_baseline"] = cpu_baseline(cfg, "fwd", budget_s=15.0)
        ref_out = [o.float() for o in fstep()[:3]]      # fp32 HIP forward of the same model / inputs (pinned to the oracle at 1e-6)

        def rank_stats(outs):
            """What a user of a reduced-precision inference mode sees: agreement of the ranking heads with the fp32 forward of
            the same (random-init) model on the same 512 samples (tests/test_mx_bench_shapes_gpu.py holds the oracle-side twin)."""
            vq, vq_ref = outs[0].float(), ref_out[0]
            top1 = float((vq.argmax(1) == vq_ref.argmax(1)).float().mean())
            in5 = float((vq.topk(5, dim=1).indices == vq_ref.argmax(1, keepdim=True)).any(1).float().mean())
            a, b = outs[2].float().view(-1), ref_out[2].view(-1)
            ra, rb = a.argsort().argsort().double(), b.argsort().argsort().double()
            rho = float(((ra - ra.mean()) * (rb - rb.mean())).sum() / ((ra - ra.mean()).norm() * (rb - rb.mean()).norm()))
            l2 = float((vq.double() - vq_ref.double()).norm() / vq_ref.double().norm())
            return {"vqa_top1_agreement": round(top1, 3), "fp32_top1_in_top5": round(in5, 3),
                    "retrieval_score_spearman": round(rho, 3), "vqa_logits_rel_l2": round(l2, 4),
                    "vs": "the fp32 forward of the same random-init model, 512 samples"}

        # round 6: the bf16 stream (bf16 tensors in HBM, v_mfma_f32_32x32x16_bf16, fp32 LayerNorm / softmax statistics) as an
        # INFERENCE mode - BASELINE.md section 4 row 2 names "fp32-MFMA (parity) / bf16" for this point
        _native.set_gemm_mode("bf16")
        try:
            fb_dt = timed(fstep, 2, n_f)
            fb_tf = 512 * n_f / fb_dt * f_total / 1e12
            tf16f, n16f = gemm_family_tf(fstep)
            extra["fwd_bf16_b512"] = {"value": round(512 * n_f / fb_dt, 2), "unit": "samples/s",
                                      "ms_per_step": round(1e3 * fb_dt / n_f, 3), "steps": n_f, "dtype": BF16_DTYPE,
                                      "speedup_vs_fp32": round(f_dt / fb_dt, 2), "model_tflops": round(fb_tf, 1),
                                      "frac_of_bf16_mfma_peak": round(fb_tf / 2500.0, 4),
                                      "rank_statistics": rank_stats(fstep()),
                                      "roofline": bf16_roofline(tf16f, n16f),
                                      "note": "the same forward on the bf16 stream of the training mode (eval, no_grad): outside the "
                                              "1e-4 parity bar by design; drift bounds tests/test_bf16_stream_gpu.py"}
        finally:
            _native.set_gemm_mode("f32")
        # BASELINE configs[4]: the same forward with the linears on quantised e4m3 operands (csrc/fp8.hip). Per-row
        # scales, fp32 accumulate / LayerNorm / attention; outside the 1e-4 bar by design (tests/test_fp8_gpu.py).
        _native.set_gemm_mode("fp8")
        try:
            f8_dt = timed(fstep, 2, n_f)
            extra["fwd_fp8_b512"] = {"value": round(512 * n_f / f8_dt, 2), "unit": "samples/s",
                                     "ms_per_step": round(1e3 * f8_dt / n_f, 3), "steps": n_f,
                                     "speedup_vs_fp32": round(f_dt / f8_dt, 2),
                                     "note": "BASELINE configs[4] direction: forward with every eligible nn.Linear on OCP "
                                             "e4m3 operands (row-wise scales, v_mfma_scale_f32_32x32x64_f8f6f4, fp32 "
                                             "accumulate); attention / LayerNorm / heads with N < 64 stay fp32"}
            f128, _, _ = forward_workload(128)
            n128 = 4 * n_f
            f128_dt = timed(f128, 2, n128)
            g128, _, _ = forward_workload(128, graph=True)
            g128_dt = timed(g128, 2, n128)
            extra["fwd_fp8_b128"] = {"value": round(128 * n128 / min(f128_dt, g128_dt), 2), "unit": "samples/s",
                                     "eager": round(128 * n128 / f128_dt, 2), "graphed": round(128 * n128 / g128_dt, 2),
                                     "ms_per_step": round(1e3 * min(f128_dt, g128_dt) / n128, 3), "steps": n128,
                                     "note": "per-GPU share of BASELINE configs[4] (batch 1024 over 8 GPUs = 128 per GPU): "
                                             "eager launches vs the forward replayed as one HIP graph "
                                             "(vilbert/graphed.py GraphedForward)"}
            del f128, g128
            # round 4: the MX (block-scaled) form of the same path - LayerNorm and the GELU epilogue emit the e4m3 codes the
            # next linear consumes, the scaled MFMA applies the block scales (csrc/mx8.hip; bars: tests/test_mx_gpu.py)
            _native.set_gemm_mode("mxfp8")
            mx_dt = timed(fstep, 2, n_f)
            mx_tf = 512 * n_f / mx_dt * f_total / 1e12
            extra["fwd_mxfp8_b512"] = {"value": round(512 * n_f / mx_dt, 2), "unit": "samples/s",
                                       "ms_per_step": round(1e3 * mx_dt / n_f, 3), "steps": n_f,
                                       "speedup_vs_fp32": round(f_dt / mx_dt, 2), "model_tflops": round(mx_tf, 1),
                                       "frac_of_fp8_mfma_peak": round(mx_tf / 5000.0, 4),
                                       "note": "BASELINE configs[4]: forward with every nn.Linear whose K and N are multiples of "
                                               "128 on MX e4m3 operands (one E8M0 scale per 32 K elements, applied by "
                                               "v_mfma_scale_f32_32x32x64_f8f6f4), codes emitted by LayerNorm / the GELU "
                                               "epilogue (no quantiser pass, no fp32 FFN activation); residual stream between the layers "
                                               "in %s, attention on bf16 q | k | v with fp32 softmax and an MX context, LayerNorm "
                                               "statistics fp32; fp8 dense MFMA peak 5000 TF" % os.environ.get("VB_MX_STREAM", "bf16"),
                                       "mx_stream": os.environ.get("VB_MX_STREAM", "bf16"),
                                       "rank_statistics": rank_stats(fstep())}
            # its own roofline object: every MX / fp8 GEMM launch of two more forwards bracketed with HIP events, single stream
            # (as the headline's); bound: the scaled-MFMA peak. Counter-side bytes of the dominant launch: tools/pmc_mx.sh
            two_mx = _vb.set_two_streams(False)
            fstep()
            torch.cuda.synchronize()
            ops.profile_linear(True)
            fstep()
            fstep()
            torch.cuda.synchronize()
            mx_ms, mx_fl, mx_n = ops.profile_linear(False)
            _vb.set_two_streams(two_mx)
            mx_roof = {"bound": "mfma", "kernel": "gemm_mx_kernel (v_mfma_scale_f32_32x32x64_f8f6f4, persistent 256x128 tiles, "
                       "8 MFMA + 2 LDS-DMA loader waves per CU) + gemm_fp8_kernel on the ragged-N heads",
                       "achieved": round(mx_fl / (mx_ms * 1e-3) / 1e12, 1) if mx_ms > 0 else 0.0, "peak": 5000.0,
                       "unit": "TFLOP/s", "frac": round(mx_fl / (mx_ms * 1e-3) / 1e12 / 5000.0, 4) if mx_ms > 0 else 0.0,
                       "launches_per_forward": mx_n // 2, "avg_launch_us": round(1e3 * mx_ms / max(mx_n, 1), 2),
                       "what": "all linear launches of 2 forwards, algorithmic 2MNK / sum of HIP-event durations (single stream)",
                       "traffic": None}
            tp = os.path.join(ROOT, "profiles", "r06_mx_gemm_traffic.json")
            if not os.path.isfile(tp):
                tp = os.path.join(ROOT, "profiles", "r04_mx_gemm_traffic.json")
            if os.path.isfile(tp):
                t0 = json.load(open(tp))["launches"][0]
                mx_roof["traffic"] = t0["hbm_bytes_corrected"]
                mx_roof["traffic_note"] = "rocprofv3 PMC, %s M=%d N=%d K=%d: %.0f MB per launch vs %.0f MB algorithmic (profiles/%s)" % (
                    t0["kernel"], t0["M"], t0["N"], t0["K"], t0["hbm_bytes_corrected"] / 1e6, t0["algorithmic_bytes"] / 1e6,
                    os.path.basename(tp))
            extra["fwd_mxfp8_b512"]["roofline"] = mx_roof
            emx, _, _ = forward_workload(128)
            emx_dt = timed(emx, 2, n128)
            gmx, _, _ = forward_workload(128, graph=True)
            gmx_dt = timed(gmx, 2, n128)
            extra["fwd_mxfp8_b128"] = {"value": round(128 * n128 / min(emx_dt, gmx_dt), 2), "unit": "samples/s",
                                       "eager": round(128 * n128 / emx_dt, 2), "graphed": round(128 * n128 / gmx_dt, 2),
                                       "ms_per_step": round(1e3 * min(emx_dt, gmx_dt) / n128, 3), "steps": n128,
                                       "note": "per-GPU share of BASELINE configs[4] (128 per GPU) in the MX mode: eager launches vs "
                                               "one HIP graph (stand-alone A/B of the same two: tools/mx_b128_ab.py, "
                                               "profiles/r04_mx_b128_ab.txt - 22.8 k eager, 34.0 k graphed)"}
            del emx, gmx
        finally:
            _native.set_gemm_mode("f32")
        del fstep, fmodel, ref_out
        torch.cuda.empty_cache()
        extra.update(large_legs())
        torch.cuda.empty_cache()
        extra.update(task_legs())
        torch.cuda.empty_cache()

    if rank == 0:
        bert_f, total_f = model_flops_per_sample(cfg, N_TOK, n_reg, "vltasks" if args.mode == "fwd" else "pretraining")
        mult = 1 if args.mode == "fwd" else 3
        exec_f = total_f
        if args.mode == "train":
            # the pre-training heads run at labelled positions only (result-identical, see
            # BertForMultiModalPreTraining._losses_at_labelled_positions): count what is executed
            H, Hv, V = cfg["hidden_size"], cfg["v_hidden_size"], cfg["vocab_size"]
            n_t = float((x["masked_lm_labels"] != -1).sum()) / B
            n_v = float((x["image_label"] == 1).sum()) / B
            exec_f = bert_f + 2 * n_t * (H * H + H * V) + 2 * n_v * (Hv * Hv + Hv * cfg["v_target_size"]) \
                + 2 * cfg["bi_hidden_size"] * 2
        sps = world * B * args.steps / elapsed
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        traffic, traffic_note = None, "not measured"
        tpath = os.path.join(ROOT, "profiles", "r05_gemm_traffic.json")
        if not os.path.isfile(tpath):
            tpath = os.path.join(ROOT, "profiles", "r03_gemm_traffic.json")
        if not os.path.isfile(tpath):
            tpath = os.path.join(ROOT, "profiles", "r02_gemm_traffic.json")
        if os.path.isfile(tpath):
            # HBM bytes per launch cannot be read from inside the process; this is the rocprofv3 PMC
            # measurement (FETCH_SIZE / WRITE_SIZE passes, gfx950 read correction) of the dominant forward
            # GEMM shape, committed under profiles/
            t0 = json.load(open(tpath))["launches"][0]
            traffic = t0["hbm_bytes_corrected"]
            traffic_note = "rocprofv3 PMC, %s M=%d N=%d K=%d: %.0f MB per launch vs %.0f MB algorithmic (profiles/%s)" % (
                t0["kernel"], t0["M"], t0["N"], t0["K"], traffic / 1e6, t0["algorithmic_bytes"] / 1e6, os.path.basename(tpath))
        line = {
            "metric": "samples/sec (%d regions, %d tokens) ViLBERT-%s %s" %
                      (N_REG, N_TOK, "base 6L/6C" if CONFIG == "bert_base_6layer_6conect.json" else CONFIG,
                       "forward" if args.mode == "fwd" else "fwd+bwd"),
            "value": round(sps, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded 36x2048 region features + 36 token ids, random-init weights)",
            "config": {"workload": "%s %s, batch %d per GPU, T=%d R=%d" %
                                   (CONFIG, "forward-only (eval, no_grad), VILBertForVLTasks incl. all heads"
                                    if args.mode == "fwd" else
                                    "train_concap step: BertForMultiModalPreTraining fwd+bwd (dropout on) + "
                                    "grad all-reduce + AdamW, %d regions + 1 global row" % N_REG, B, N_TOK, n_reg),
                       "per_gpu_batch": B, "global_batch": B * world,
                       "parallelism": "dp%d" % world, "visible_gpus": torch.cuda.device_count(),
                       "launch": "one HIP graph per step" if args.graph else "eager",
                       "gflop_per_sample_model": round(mult * total_f / 1e9, 3),
                       "gflop_per_sample_executed": round(mult * exec_f / 1e9, 3),
                       "gflop_per_sample_bertmodel": round(mult * bert_f / 1e9, 3)},
            "model_tflops": round(sps / world * mult * exec_f / 1e12, 2),
            "model_frac_of_fp32_mfma_peak": round(sps / world * mult * exec_f / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
            "roofline": {"bound": "mfma", "kernel": "fp32 GEMM family on v_mfma_f32_16x16x4_f32: gemm_v4_kernel / gemm_v4w_kernel "
                                                    "(persistent, one 13-wave block per CU: text-stream forward, dgrad, FFN "
                                                    "wgrad) + gemm_v2_kernel (4-wave blocks: 37-region image stream, remaining "
                                                    "wgrad); ragged launches gemm_f32_kernel (v_mfma_f32_32x32x2_f32)",
                         "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "traffic_note": traffic_note,
                         "what": "all GEMM launches of %d extra step(s) (fwd%s), algorithmic 2MNK FLOPs / "
                                 "sum of HIP-event durations (profiled step single-stream)" % (prof_steps, "" if args.mode == "fwd" else " + dgrad + wgrad"),
                         "launches_per_step": gemm_launches // prof_steps,
                         "avg_launch_us": round(1e3 * gemm_ms / max(gemm_launches, 1), 2),
                         "flops_per_launch_avg": round(gemm_flops / max(gemm_launches, 1), 0)},
        }
        line["config"]["gemm_mode"] = args.gemm_mode
        if args.mode == "train":
            line["config"]["label_gather"] = label_gather_info
        # the STATE, not the flag: ordered split-K is the default (VB_DETERMINISTIC=0 switches to atomics); fallbacks =
        # split launches that wanted the ordered reduce and ran with atomics (no free workspace slice)
        line["config"]["deterministic_wgrad"] = bool(_native._DET["wanted"]) and _native.deterministic_workspace(device) is not None
        line["config"]["deterministic_fallbacks"] = _native.deterministic_fallbacks()
        line.update(extra)
        if alt:
            if "bf16" in alt:
                alt["bf16"]["model_tflops"] = round(alt["bf16"]["value"] * mult * exec_f / 1e12, 1)
                alt["bf16"]["model_frac_of_bf16_mfma_peak"] = round(alt["bf16"]["model_tflops"] / 2500.0, 4)
            line["alt_gemm_modes"] = alt
        if host_leg is not None:
            line["host_inputs"] = host_leg
        if args.gemm_mode in ("fp8", "mxfp8"):
            line["dtype"] = "OCP e4m3 operands, fp32 accumulate, forward linears only - NOT inside the 1e-4 parity bar; " + (
                "one scale per row (csrc/fp8.hip; drift: tests/test_fp8_gpu.py)" if args.gemm_mode == "fp8" else
                "MX block format: one E8M0 scale per 32 contraction elements applied by the scaled MFMA, bf16 residual stream "
                "(VB_MX_STREAM=%s), bf16 attention operands with fp32 softmax (csrc/mx8.hip; drift: tests/test_mx_gpu.py, "
                "test_mx_bench_shapes_gpu.py)" % os.environ.get("VB_MX_STREAM", "bf16"))
            line["roofline"].update(peak=5000.0, frac=round(achieved / 5000.0, 4),
                                    kernel="gemm_fp8_kernel (v_mfma_scale_f32_32x32x64_f8f6f4) - meaningful with --mode "
                                           "forward (backward GEMMs stay fp32)",
                                    peak_note="fp8 dense MFMA peak 5000 TF")
        elif args.gemm_mode != "f32":
            peak = 2500.0 / {"bf16x6": 6, "bf16x3": 3, "bf16": 1}[args.gemm_mode]
            line["dtype"] = "f32 operands split into bf16 planes (%s), fp32 accumulate" % args.gemm_mode
            line["roofline"].update(peak=round(peak, 1), frac=round(achieved / peak, 4),
                                    kernel="gemm_planes_kernel (v_mfma_f32_32x32x16_bf16, %s)" % args.gemm_mode,
                                    peak_note="bf16 dense MFMA peak 2500 TF / MFMA products per fp32 product")
            if args.gemm_mode == "bf16":
                line["dtype"] = BF16_DTYPE
                line["roofline"].update(bf16_roofline(achieved, gemm_launches // prof_steps))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, args.mode)
        # the legs a reader needs first, again, as the LAST key: the driver keeps the final 8 KB of stdout
        def pick(d, *keys):
            for k in keys:
                d = d.get(k) if isinstance(d, dict) else None
            return d
        summ = {"headline_samples_per_s": line["value"], "headline_ms_per_step": line["ms_per_step"],
                "headline_roofline_frac": line["roofline"]["frac"],
                "b64": pick(line, "b64", "value"), "global512_one_gpu": pick(line, "global512", "value"),
                "fwd_b512": pick(line, "fwd_b512", "value"), "fwd_b512_frac": pick(line, "fwd_b512", "frac_of_fp32_mfma_peak"),
                "fwd_bf16_b512": pick(line, "fwd_bf16_b512", "value"),
                "fwd_bf16_b512_gemm_frac": pick(line, "fwd_bf16_b512", "roofline", "frac"),
                "fwd_bf16_b512_rank": pick(line, "fwd_bf16_b512", "rank_statistics"),
                "fwd_mxfp8_b512": pick(line, "fwd_mxfp8_b512", "value"),
                "fwd_mxfp8_b512_gemm_frac": pick(line, "fwd_mxfp8_b512", "roofline", "frac"),
                "fwd_mxfp8_b512_rank": pick(line, "fwd_mxfp8_b512", "rank_statistics"),
                "train_bf16x6_b256": pick(line, "alt_gemm_modes", "bf16x6", "value"),
                "train_bf16x6_note": "fp32 operands as 3 bf16 planes, 6 MFMA products, fp32 accumulate: passes the SAME 1e-4 parity "
                                     "tests against the reference's goldens (tests/test_gemm_modes_gpu.py); opt-in, not the headline",
                "train_bf16_b256": pick(line, "alt_gemm_modes", "bf16", "value"),
                "train_bf16_gemm_frac": pick(line, "alt_gemm_modes", "bf16", "roofline", "frac"),
                "train_bf16_b64": pick(line, "alt_gemm_modes", "bf16", "b64", "value"),
                "train_bf16_b64_eager": pick(line, "alt_gemm_modes", "bf16", "b64", "eager"),
                "train_bf16_b64_graphed": pick(line, "alt_gemm_modes", "bf16", "b64", "graphed")}
        pred = pick(line, "comm_model", "buckets_64_mib", "predicted_global512_n8")
        g512 = summ["global512_one_gpu"]
        if pred and g512:
            summ["n8_model_fp32"] = {"samples_per_s_ring": pred["samples_per_s_ring"], "samples_per_s_direct": pred["samples_per_s_direct"],
                                     "x_one_gpu_ring": round(pred["samples_per_s_ring"] / g512, 2),
                                     "x_one_gpu_direct": round(pred["samples_per_s_direct"] / g512, 2),
                                     "what": "MODEL of 8 x 64 samples (measured one-GPU step at 64 + exposed exchange) / measured "
                                             "one-GPU rate at global batch 512"}
        pred16 = pick(line, "alt_gemm_modes", "bf16", "b64", "comm_model", "buckets_64_mib", "predicted_global512_n8")
        if pred16 and summ["train_bf16_b256"]:
            summ["n8_model_bf16"] = {"samples_per_s_ring": pred16["samples_per_s_ring"],
                                     "samples_per_s_direct": pred16["samples_per_s_direct"],
                                     "x_one_gpu_b256_direct": round(pred16["samples_per_s_direct"] / summ["train_bf16_b256"], 2),
                                     "what": "MODEL, bf16 buckets: 8 x 64 samples / the measured one-GPU bf16 rate at batch 256"}
        line["summary"] = summ
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio; flush it first so the JSON is the LAST line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
