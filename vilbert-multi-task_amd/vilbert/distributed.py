"""Data-parallel gradient exchange for one-process-per-GPU training (RCCL over xGMI).

Replaces what the reference gets from ``apex.parallel.DistributedDataParallel`` (train_concap.py:506-513
overlapped / bucketed; train_tasks.py:490-497 ``delay_allreduce=True``): the only collective on the hot
path is one gradient all-reduce (average) per optimizer step.

Design for MI355X (8 GPUs, 7 xGMI links per GPU, 288 GB HBM each):
  * ZERO COPY: the buckets are contiguous ranges of the model's gradient arena (arena.py). The backward kernels
    (wgrad GEMM epilogues, LayerNorm / embedding gradient kernels) write every gradient at its final address inside
    its bucket, ``param.grad`` is a view of it, the all-reduce runs in place and the optimizer reads the same memory:
    no pack copy before the collective, no copy back after it (gradients produced by foreign autograd nodes are
    copied in, the exception);
  * LARGE buckets (default 64 MiB = 16 M fp32 elements: a ~1 GB model is 14 collectives, each big enough to sit on
    RCCL's bandwidth plateau over xGMI, few enough that launch latency is noise; rounds 1-2 used 256 MiB, but then the
    first all-reduce starts only after 55 % of backward and the last 190 MB bucket is fully exposed - bench.py's
    comm_model, measured launch times + the xGMI cost model: 1.6 ms exposed instead of 3.4 ms at 64 samples per GPU),
    laid out in reverse parameter-registration order,
    i.e. roughly the order backward produces gradients (text layer 11, image layer 5, connection 5, ... embeddings
    last); each bucket's all-reduce is issued asynchronously the moment its last expected gradient arrives, so the
    communication of bucket i overlaps the backward GEMMs of bucket i + 1;
  * the set of parameters that receive a gradient is learnt per bucket on the first pass (``biOutput.q_dense1/2`` never
    do; most task heads do not under train_tasks) and tracked as a SET: a bucket is reduced early only when exactly
    the expected parameters have arrived; a gradient that shows up for a parameter outside that set before the launch
    simply extends the wait, one that shows up AFTER its bucket was reduced raises (silently averaging a stale slice
    would corrupt training) - use ``delay_allreduce=True`` (reduce everything after backward, the reference's choice
    for multi-task training) when the used set changes from step to step.
  * TWO exchange algorithms per bucket (``algorithm=``): ``"ring"`` = one ``all_reduce`` (RCCL picks ring / tree: on
    the 8-GPU xGMI mesh a ring moves 2 (N-1)/N S bytes over ONE link per GPU, ~153 GB/s); ``"direct"`` = the two-phase
    form SURVEY.md section 5 / 7.1-8 asks for - ``reduce_scatter_tensor`` of the arena range into this rank's 1/N shard
    (in place: the shard is a view of the range), then ``all_gather_into_tensor`` of the shards back into the range;
    with every GPU directly linked to the other seven, each phase sends S/N bytes to each peer over ITS OWN link, i.e.
    2 S/N per link instead of 2 (N-1)/N S - the arithmetic bench.py's comm_model prices. Both produce the same average
    (tests/test_distributed_cpu.py compares them bit for bit on gloo); which one is faster on a given node is a
    measurement (`bench.py --gpus N --ddp-algorithm direct|ring`). Arena slices are padded to a multiple of
    lcm(4, world) elements so that every bucket divides evenly into shards;
  * optional reduced-precision exchange (``bucket_dtype=torch.bfloat16``, for the opt-in bf16 / fp8 GEMM modes whose
    gradients carry ~3 significant digits anyway): the bucket is rounded into a bf16 staging buffer, exchanged at half
    the bytes, and widened back into the fp32 arena range; master gradients, optimizer state and weights stay fp32.
Works with any ``torch.distributed`` backend (``nccl`` = RCCL on ROCm; ``gloo`` for the CPU tests).
"""
import contextlib
import math
import os

import torch
import torch.distributed as dist
from torch import nn

from . import _native as _N
from .arena import GradArena

_LAUNCH_MODE = os.environ.get("VB_DDP_LAUNCH", "cur")      # "ls": bucket exchanges issued from a launch stream (experiment, see _launch)


class _Bucket(object):
    def __init__(self, arena, first, last):
        """parameters arena.params[first:last] -> flat[lo:hi]"""
        self.first, self.last = first, last
        self.lo = arena.offsets[first]
        al = arena.align_elems
        self.hi = arena.offsets[last - 1] + (arena.params[last - 1].numel() + al - 1) // al * al
        self.flat = arena.flat[self.lo:self.hi]
        self.stage = None        # reduced-precision staging copy of the range (bucket_dtype)
        self.expected = None     # set of parameter indices that received a gradient in the previous pass
        self.index = -1          # position in the wrapper's bucket list (launch order = reverse registration order)
        self.reset()

    def reset(self):
        self.ready = set()
        self.hits = 0            # how many of `expected` are in `ready` (ready == expected <=> hits == len(expected) == len(ready))
        self.streams = {}        # raw stream handle -> stream on which some gradient of this bucket was produced
        self.work = None
        self.launched = False


class DistributedDataParallel(nn.Module):
    """``DistributedDataParallel(model[, delay_allreduce=False, message_size=...])`` - apex-compatible
    constructor subset. ``model.module`` is the wrapped network (the reference checks ``hasattr(model,
    "module")`` when saving, train_concap.py:662-664)."""

    def __init__(self, module, delay_allreduce=False, message_size=16 * 1024 * 1024, process_group=None,
                 algorithm=None, bucket_dtype=None, direct_at_world_size_one=None, **_unused):
        super(DistributedDataParallel, self).__init__()
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed.init_process_group must be called first")
        import os
        from . import _native as N
        # Reduced-precision training (the bf16 mode: process-wide `set_gemm_mode("bf16")` or `model.half()`, the reference's
        # train_concap.py:504-513 order - half() BEFORE the DDP wrapper) changes the defaults (round 6): its step is 3.7x
        # shorter than the fp32 step while the gradient message is the same 1 GB, so a ring all-reduce of fp32 buckets
        # (11.4 ms at 8 GPUs) no longer hides behind backward - bf16 buckets over the direct exchange move
        # 2 (S/2)/N bytes per link (0.8 ms). Explicit arguments / VB_DDP_ALGORITHM still win; bucket_dtype=torch.float32
        # keeps fp32 buckets in that mode.
        reduced = bool(N.bf16_stream() or getattr(module, "_vb_bf16", False))
        algorithm = algorithm or os.environ.get("VB_DDP_ALGORITHM") or ("direct" if reduced else "ring")
        if bucket_dtype is None and reduced and os.environ.get("VB_DDP_BUCKET_DTYPE", "bf16") == "bf16":
            bucket_dtype = torch.bfloat16
        if algorithm not in ("ring", "direct"):
            raise ValueError("algorithm must be 'ring' (one all_reduce per bucket) or 'direct' (reduce_scatter + "
                             "all_gather), got %r" % (algorithm,))
        if bucket_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("bucket_dtype must be None / torch.float32 / torch.bfloat16")
        self.algorithm = algorithm
        # A group of ONE rank normally takes the plain all_reduce (nothing to exchange). direct_at_world_size_one (or
        # VB_DDP_FORCE_DIRECT=1) sends it through the in-place reduce_scatter_tensor(shard of buf) + all_gather_into_tensor pair
        # anyway, so that the two-phase exchange runs on RCCL on a single-GPU box (tests/test_arena_gpu.py, bench.py --force-ddp
        # --ddp-algorithm direct); results are unchanged.
        self.direct_at_world_size_one = bool(int(os.environ.get("VB_DDP_FORCE_DIRECT", "0"))) \
            if direct_at_world_size_one is None else bool(direct_at_world_size_one)
        self.bucket_dtype = None if bucket_dtype in (None, torch.float32) else bucket_dtype
        self.module = module
        self.delay_allreduce = delay_allreduce
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.bucket_elems = int(message_size)
        self._native_avg = dist.get_backend(process_group) == "nccl"

        # identical start on every rank (apex broadcasts from rank 0 in its constructor as well)
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=0, group=process_group)

        seen, params = set(), []
        for p in module.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        # gradient arena in reverse registration order; the buckets are consecutive runs of it
        # (slices padded to lcm(4, world) elements: 16-byte alignment for the kernels, whole shards for "direct")
        self.arena = GradArena(list(reversed(params)), align_elems=4 * self.world_size // math.gcd(4, self.world_size))
        self._buckets, first, elems = [], 0, 0
        for i, p in enumerate(self.arena.params):
            elems += p.numel()
            if elems >= self.bucket_elems:
                self._buckets.append(_Bucket(self.arena, first, i + 1))
                first, elems = i + 1, 0
        if first < len(self.arena.params):
            self._buckets.append(_Bucket(self.arena, first, len(self.arena.params)))
        for k, b in enumerate(self._buckets):
            b.index = k
        # measurement hook (bench.py comm_model): a list here receives (bucket index, bytes, timing event recorded on the
        # launching stream at the moment the bucket's all-reduce is enqueued) for every launch
        self.trace = None
        self._where = {}
        for b in self._buckets:
            for i in range(b.first, b.last):
                p = self.arena.params[i]
                self._where[id(p)] = (b, i)
                p.register_post_accumulate_grad_hook(self._hook)
        self.arena.add_listener(self)
        self._view_ptrs = [v.data_ptr() for v in self.arena.views]
        self._cuda = self.arena.flat.is_cuda
        self._dev_index = self.arena.flat.device.index if self._cuda else None
        self._launch_stream = None  # the stream bucket exchanges are issued from (see _launch)
        self._pass = None           # autograd graph-task id of the pass being tracked
        self._finalized = True

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    # ---- backward-time machinery -------------------------------------------------------------
    def _enter_pass(self):
        """Called from the first gradient event of a backward pass (keyed on autograd's graph-task id, so an aborted
        backward cannot leave the state half-way)."""
        task = torch._C._current_graph_task_id()
        if task != self._pass or self._finalized:
            self._pass, self._finalized = task, False
            for b in self._buckets:
                b.reset()
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def _hook(self, param):
        # (538 calls per backward pass: kept lean - cached slice addresses, raw stream handles, a counter instead of a set
        # comparison; in the bf16 mode, whose step the host can barely keep ahead of, the hooks + the finalize loop were
        # ~4.8 ms of a 26 ms step, tools/ddp_overhead.py)
        self._enter_pass()
        b, i = self._where[id(param)]
        g = param.grad
        if g is not None and g.data_ptr() != self._view_ptrs[i]:
            view = self.arena.views[i]
            # produced by a foreign autograd node: move it into the bucket (the native kernels write in place).
            # The arena's pass must have begun first: a foreign gradient can arrive before the first native claim()
            # (a torch head on top of the native body), and the fill of that claim would otherwise wipe the copy.
            self.arena.note_foreign_write(i)
            view.copy_(g)
        if b.launched:
            raise RuntimeError(
                "DistributedDataParallel: parameter of shape %s received its gradient after its bucket had been "
                "all-reduced - the set of used parameters changed between steps; construct with "
                "delay_allreduce=True (the reference's multi-task mode, train_tasks.py:497)" % (tuple(param.shape),))
        if i not in b.ready:
            b.ready.add(i)
            if b.expected is not None and i in b.expected:
                b.hits += 1
        if self._cuda:
            # the model runs its text / image streams on two HIP streams, so gradients of one bucket
            # are produced on different streams: remember which (no event per gradient - a stream is
            # in-order, so waiting for the stream at launch time covers every gradient enqueued on it)
            handle = _N.raw_stream(self._dev_index)
            if handle not in b.streams:
                b.streams[handle] = torch.cuda.current_stream(self.arena.flat.device)
        if not self.delay_allreduce and b.expected is not None and b.hits == len(b.expected) and len(b.ready) == b.hits:
            self._launch(b)

    def _launch(self, b):
        """Start the in-place all-reduce of the bucket's arena range."""
        from . import autograd_ops as _A
        # What the exchange has to wait for - the other encoder stream, the weight-gradient side streams - is waited for by
        # the CURRENT backward stream; the process group's stream orders itself behind that stream at the call.
        # (Round 6 experiment, VB_DDP_LAUNCH=ls: issue the waits and the collective from a launch stream of the wrapper
        # instead, so that the backward stream never waits for the side streams. Measured at world size 1 with the collectives
        # stubbed out (tools/ddp_overhead2.py, bf16 step 26.4 ms): waits on the current stream + 1.6 ms, launch stream
        # + 10.9 ms - a third stream that waits on events of the compute stream 14 times per pass costs far more than the
        # waits it takes off that stream; profiles/r06_ddp_overhead_launch_modes.txt. Not the default.)
        cuda = b.flat.is_cuda
        ctx = contextlib.nullcontext()
        if cuda and _LAUNCH_MODE == "ls":
            cur = torch.cuda.current_stream(b.flat.device)
            ls = self._launch_stream
            if ls is None or ls.device != b.flat.device:
                ls = self._launch_stream = torch.cuda.Stream(device=b.flat.device)
            ls.wait_stream(cur)
            _A.join_wgrad_streams(into=ls)
            for handle, st in b.streams.items():
                if handle != cur.cuda_stream:
                    ls.wait_stream(st)
            ctx = torch.cuda.stream(ls)
        elif cuda:
            cur = torch.cuda.current_stream(b.flat.device)
            _A.join_wgrad_streams()      # weight gradients are written on side streams of the backward streams
            for handle, st in b.streams.items():
                if handle != cur.cuda_stream:
                    cur.wait_stream(st)
        with ctx:
            if self.trace is not None and cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                self.trace.append((b.index, b.flat.numel() * b.flat.element_size(), ev))
            op = dist.ReduceOp.AVG if self._native_avg else dist.ReduceOp.SUM
            buf = b.flat
            if self.bucket_dtype is not None:
                if b.stage is None:
                    b.stage = torch.empty(b.flat.numel(), dtype=self.bucket_dtype, device=b.flat.device)
                b.stage.copy_(b.flat)                 # round to the exchange precision (one elementwise pass)
                buf = b.stage
            if self.algorithm == "ring" or (self.world_size == 1 and not self.direct_at_world_size_one):
                b.work = [dist.all_reduce(buf, op=op, group=self.group, async_op=True)]
            else:
                # two-phase exchange, both phases in place: this rank's shard is a view of the range
                n = buf.numel() // self.world_size
                r = dist.get_rank(self.group)
                shard = buf[r * n:(r + 1) * n]
                rs = dist.reduce_scatter_tensor(shard, buf, op=op, group=self.group, async_op=True)
                if not self._native_avg:
                    rs.wait()                         # gloo runs asynchronous work on a thread pool: order the phases by hand
                # (RCCL enqueues both on the process group's stream, in issue order)
                ag = dist.all_gather_into_tensor(buf, shard, group=self.group, async_op=True)
                b.work = [rs, ag]
        b.launched = True

    def arena_backward_done(self):
        """Arena listener: the backward pass that wrote gradients is over (covers passes in which no
        post-accumulate hook fired, e.g. pure accumulation into existing gradients)."""
        if not self._finalized and self._pass is not None:
            self._finalize()

    def _finalize(self):
        if self._finalized:
            return
        self._finalized = True
        for b in self._buckets:  # whatever did not complete during backward (unused params, delayed mode)
            if not b.launched:
                self._launch(b)
        for b in self._buckets:
            for w in b.work:
                w.wait()
            if b.stage is not None:
                b.flat.copy_(b.stage)             # widen the exchanged values back into the fp32 arena range
            if not self._native_avg:
                b.flat.div_(self.world_size)
            params, ptrs = self.arena.params, self._view_ptrs
            for i in b.ready:
                g = params[i].grad
                if g is None or g.data_ptr() != ptrs[i]:
                    params[i].grad = self.arena.alias(i)
            b.expected = set(b.ready)    # learnt / refreshed for the next pass
