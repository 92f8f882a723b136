"""Device-side input pipeline of the pre-training step.

The reference finishes every batch on the host (numpy: global mean-region row, concatenations -
vilbert/datasets/concept_cap_dataset.py:241-282), converts to torch, copies ten tensors with ``.cuda()``
inside the step and edits the labels for objective 1 with a few torch ops (train_concap.py:529-540). At
batch 512 that is ~270 MB of numpy concatenation per step on one core - comparable to the GPU step itself.

Here the RAW worker arrays are staged through pinned host buffers, copied on a dedicated HIP stream while
the previous step computes (double buffered), and finished by one native pass (csrc/batch.hip,
``vb_concap_finish_batch``). ``DeviceBatchPipeline`` yields exactly the tuple the reference training loop
unpacks after its own ``.cuda()`` calls and label edit:
    (input_ids, input_mask, segment_ids, lm_label_ids, is_next, image_feat, image_loc, image_target,
     image_label, image_mask)
"""
import ctypes
import os

import numpy as np
import torch

from . import _native as N

# order of the raw tuple produced by the dataset workers (concept_cap_dataset.py:243-245)
RAW_FIELDS = ("input_ids", "input_mask", "segment_ids", "lm_label_ids", "is_next", "image_feat", "image_loc",
              "image_target", "image_label", "image_mask", "masked_label")
_DTYPES = dict(input_ids=torch.int64, input_mask=torch.int64, segment_ids=torch.int64, lm_label_ids=torch.int64,
               is_next=torch.int64, image_feat=torch.float32, image_loc=torch.float32, image_target=torch.float32,
               image_label=torch.int64, image_mask=torch.int64, masked_label=torch.int64)


def finish_batch(raw, objective=0):
    """raw: dict of DEVICE tensors named as RAW_FIELDS (worker output, unchanged). Returns the 10-tuple the
    training loop feeds to the model. One native launch pair on the current stream; no torch arithmetic."""
    feat = raw["image_feat"]
    if feat.dim() != 3:
        raise RuntimeError("image_feat must be [batch, regions, feat_dim]")
    B, R, F = feat.shape
    T = raw["lm_label_ids"].shape[1]
    dev = feat.device
    out_feat = torch.empty(B, R + 1, F, dtype=torch.float32, device=dev)
    out_loc = torch.empty(B, R + 1, 5, dtype=torch.float32, device=dev)
    out_mask = torch.empty(B, R + 1, dtype=torch.int64, device=dev)
    out_ilab = torch.empty(B, R, dtype=torch.int64, device=dev)
    out_lm = torch.empty(B, T, dtype=torch.int64, device=dev)
    a = N.ConcapBatch()
    a.batch, a.regions, a.tokens, a.feat_dim, a.objective = B, R, T, F, int(objective)

    def f32(name, shape):
        t = raw[name]
        if tuple(t.shape) != shape or not t.is_contiguous():
            raise RuntimeError("%s: expected a contiguous tensor of shape %s, got %s" % (name, shape, tuple(t.shape)))
        return N.dev_f32(t, name)

    def i64(name, shape):
        t = raw[name]
        if tuple(t.shape) != shape:
            raise RuntimeError("%s: expected shape %s, got %s" % (name, shape, tuple(t.shape)))
        return N.dev_i64(t, name)

    a.image_feat, a.image_loc = f32("image_feat", (B, R, F)), f32("image_loc", (B, R, 5))
    a.image_mask, a.masked_label = i64("image_mask", (B, R)), i64("masked_label", (B, R))
    a.is_next, a.image_label = i64("is_next", (B,)), i64("image_label", (B, R))
    a.lm_label_ids = i64("lm_label_ids", (B, T))
    a.out_image_feat, a.out_image_loc = out_feat.data_ptr(), out_loc.data_ptr()
    a.out_image_mask, a.out_image_label, a.out_lm_label_ids = out_mask.data_ptr(), out_ilab.data_ptr(), out_lm.data_ptr()
    N.check(N.lib().vb_concap_finish_batch(N.stream_ptr(), ctypes.byref(a)), "vb_concap_finish_batch")
    return (raw["input_ids"], raw["input_mask"], raw["segment_ids"], out_lm, raw["is_next"], out_feat, out_loc,
            raw["image_target"], out_ilab, out_mask)


class _Slot(object):
    """One pinned host staging set + its device twin. ``copied``: the H2D copy of the current contents has
    finished (the host may refill the pinned buffers; compute waits on it). ``released``: the step that read
    the device buffers has run (the copy stream waits on it before overwriting them)."""

    def __init__(self):
        self.host, self.dev, self.copied, self.released, self.extra = {}, {}, None, None, ()

    def ensure(self, name, shape, dtype, device):
        h = self.host.get(name)
        if h is None or tuple(h.shape) != tuple(shape):
            self.host[name] = torch.empty(shape, dtype=dtype).pin_memory()
            self.dev[name] = torch.empty(shape, dtype=dtype, device=device)
        return self.host[name], self.dev[name]


class DeviceBatchPipeline(object):
    """Iterate over ``source`` (an iterable of RAW worker batches: tuples of numpy arrays in RAW_FIELDS order,
    optionally followed by extra items such as image ids - what ``self.ds.get_data()`` yields inside the
    reference loader) and yield finished device batches (+ the extra items).

    Batch i+1 is staged on a dedicated stream right after the caller has enqueued step i, so the transfer runs
    under step i. Two staging modes: "direct" (default; the runtime copies straight from the pageable numpy
    memory) and "pinned" (numpy -> pinned buffer -> asynchronous DMA; on this platform the CPU's writes into
    the pinned mapping are the slow part).
    (Staging from a background thread was measured and is slower: the step's ~1500 kernel launches are
    Python-side work and lose more to GIL hand-offs than the overlap wins - 161 vs 137 ms per step at batch
    256.) The id / mask / target tensors of a yielded batch alias the slot's device buffers: they stay valid
    until the caller asks for batch i + depth - 1 (do not keep them across iterations)."""

    def __init__(self, source, device, objective=0, depth=2, staging=None):
        self.source, self.device, self.objective = source, torch.device(device), int(objective)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = [_Slot() for _ in range(max(2, depth))]
        # "pinned": numpy -> pinned host buffer -> asynchronous DMA.  "direct": hand the pageable numpy memory to
        # the runtime's own copy path (it stages through its internal pinned chunks; blocks the host for the
        # duration of the transfer but never makes the CPU write through an uncached host mapping).
        self.staging = staging or os.environ.get("VB_PIPE_STAGING", "direct")   # measured: 131 vs 137-154 ms / step

    def _stage(self, slot, batch):
        if slot.copied is not None:
            slot.copied.synchronize()                       # pinned buffers free again (long done)
        if slot.released is not None:
            self.copy_stream.wait_event(slot.released)      # device buffers free once that step has run
        fields = dict(zip(RAW_FIELDS, batch[:len(RAW_FIELDS)]))
        slot.extra = tuple(batch[len(RAW_FIELDS):])
        with torch.cuda.stream(self.copy_stream):
            for name in RAW_FIELDS:
                arr = np.ascontiguousarray(fields[name])
                h, d = slot.ensure(name, arr.shape, _DTYPES[name], self.device)
                if self.staging == "direct" and arr.dtype == h.numpy().dtype:
                    d.copy_(torch.from_numpy(arr), non_blocking=False)
                else:
                    h.copy_(torch.from_numpy(arr))          # dtype conversion (e.g. int32 ids) happens here
                    d.copy_(h, non_blocking=True)
            slot.copied = torch.cuda.Event()
            slot.copied.record(self.copy_stream)

    def __iter__(self):
        it = iter(self.source)
        batch = next(it, None)
        if batch is None:
            return
        idx = 0
        self._stage(self.slots[0], batch)
        while True:
            slot = self.slots[idx]
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(slot.copied)
            for t in slot.dev.values():
                t.record_stream(cur)      # allocated on the copy stream, read by kernels of the compute stream
            out = finish_batch(slot.dev, self.objective)
            yield out + slot.extra
            slot.released = torch.cuda.Event()
            slot.released.record(torch.cuda.current_stream(self.device))   # after the caller's step
            batch = next(it, None)
            if batch is None:
                return
            idx = (idx + 1) % len(self.slots)
            self._stage(self.slots[idx], batch)
