"""Data-parallel gradient exchange for one-process-per-GPU training (RCCL over xGMI).

Replaces what the reference gets from ``apex.parallel.DistributedDataParallel`` (train_concap.py:506-513
overlapped / bucketed; train_tasks.py:490-497 ``delay_allreduce=True``): the only collective on the hot
path is one gradient all-reduce (average) per optimizer step.

Design for MI355X (8 GPUs, 7 xGMI links per GPU, 288 GB HBM each):
  * gradients live in a few LARGE flat fp32 buckets (default 256 MiB: a ~1 GB model is 4-5 collectives,
    big enough to sit on RCCL's bandwidth plateau over xGMI, few enough that launch latency is noise);
  * buckets are filled in reverse parameter-registration order, i.e. roughly the order backward produces
    gradients (text layer 11, image layer 5, connection 5, ... embeddings last), and each bucket's
    all-reduce is issued asynchronously the moment its last gradient arrives, so communication of
    bucket i overlaps the backward GEMMs of bucket i+1;
  * parameters that never receive a gradient (``biOutput.q_dense1/2`` always; most task heads under
    train_tasks) are learnt on the first step and no longer block their bucket; with
    ``delay_allreduce=True`` nothing is assumed and every bucket is reduced after backward (the
    reference's choice for multi-task training, where the unused set changes per task);
  * after the reduce ``param.grad`` is a VIEW into the bucket: no copy back.
Works with any ``torch.distributed`` backend (``nccl`` = RCCL on ROCm; ``gloo`` for the CPU tests).
"""
import torch
import torch.distributed as dist
from torch import nn


class _Bucket(object):
    def __init__(self, params, device, dtype):
        self.params = params
        self.offsets, total = [], 0
        for p in params:
            self.offsets.append(total)
            total += (p.numel() + 3) // 4 * 4  # keep every slice 16-byte aligned
        self.flat = torch.zeros(total, device=device, dtype=dtype)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(params, self.offsets)]
        self.expected = len(params)
        self.reset()

    def reset(self):
        self.ready = set()
        self.streams = {}    # raw stream handle -> stream on which some gradient of this bucket was produced
        self.work = None
        self.launched = False


class DistributedDataParallel(nn.Module):
    """``DistributedDataParallel(model[, delay_allreduce=False, message_size=...])`` - apex-compatible
    constructor subset. ``model.module`` is the wrapped network (the reference checks ``hasattr(model,
    "module")`` when saving, train_concap.py:662-664)."""

    def __init__(self, module, delay_allreduce=False, message_size=64 * 1024 * 1024, process_group=None, **_unused):
        super(DistributedDataParallel, self).__init__()
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed.init_process_group must be called first")
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
        self._buckets, cur, cur_elems = [], [], 0
        for p in reversed(params):
            cur.append(p)
            cur_elems += p.numel()
            if cur_elems >= self.bucket_elems:
                self._buckets.append(_Bucket(cur, p.device, p.dtype))
                cur, cur_elems = [], 0
        if cur:
            self._buckets.append(_Bucket(cur, cur[0].device, cur[0].dtype))
        self._where = {}
        for b in self._buckets:
            for i, p in enumerate(b.params):
                self._where[id(p)] = (b, i)
                p.register_post_accumulate_grad_hook(self._make_hook(p))
        self._callback_queued = False
        self._never_used = None  # learnt on the first backward (ids of params without a gradient)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    # ---- backward-time machinery -------------------------------------------------------------
    def _make_hook(self, p):
        def hook(param):
            if not self._callback_queued:
                self._callback_queued = True
                for b in self._buckets:
                    b.reset()
                torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
            b, i = self._where[id(param)]
            b.ready.add(i)
            if param.grad is not None and param.grad.is_cuda:
                # the model runs its text / image streams on two HIP streams, so gradients of one bucket
                # are produced on different streams: remember which (no event per gradient - a stream is
                # in-order, so waiting for the stream at pack time covers every gradient enqueued on it)
                st = torch.cuda.current_stream(param.grad.device)
                b.streams.setdefault(st.cuda_stream, st)
            if not self.delay_allreduce and self._never_used is not None and not b.launched \
                    and len(b.ready) >= b.expected:
                self._launch(b)
        return hook

    def _launch(self, b):
        """Pack the bucket (one multi-tensor copy; unused slices are zero) and start its all-reduce."""
        if b.streams:
            cur = torch.cuda.current_stream()
            for handle, st in b.streams.items():
                if handle != cur.cuda_stream:
                    cur.wait_stream(st)
        src, dst = [], []
        for i, p in enumerate(b.params):
            if p.grad is not None and p.grad.data_ptr() != b.views[i].data_ptr():
                src.append(p.grad)
                dst.append(b.views[i])
            elif p.grad is None:
                b.views[i].zero_()
        if src:
            torch._foreach_copy_(dst, src)
        op = dist.ReduceOp.AVG if self._native_avg else dist.ReduceOp.SUM
        b.work = dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)
        b.launched = True

    def _finalize(self):
        self._callback_queued = False
        for b in self._buckets:  # whatever did not complete during backward (unused params, delayed mode)
            if not b.launched:
                self._launch(b)
        for b in self._buckets:
            b.work.wait()
            if not self._native_avg:
                b.flat.div_(self.world_size)
            for i, p in enumerate(b.params):
                if i in b.ready:
                    p.grad = b.views[i]
        if self._never_used is None:
            self._never_used = set()
            for b in self._buckets:
                unused = [i for i in range(len(b.params)) if i not in b.ready]
                self._never_used.update(id(b.params[i]) for i in unused)
                b.expected = len(b.params) - len(unused)
