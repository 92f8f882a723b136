"""Gradient arena: every parameter gradient of a model lives at a FIXED address inside one flat fp32 buffer.

Why (MI355X, 288 GB HBM - the 1 GB of gradients is nothing, its traffic and launch count are):
  * the wgrad GEMM (split-K atomics), the fused bias-gradient column sums and the embedding scatter all ADD into
    zero-filled memory: with per-call ``torch.zeros`` targets that was ~250 fill launches + 1 GB of allocator churn
    per step; the arena is zero-filled by ONE memset at the first gradient write of a backward pass;
  * ``param.grad`` becomes a view of the arena (autograd "steals" the tensor a backward node returns, so returning a
    fresh alias of the arena slice costs no copy): the data-parallel buckets (distributed.py) ARE arena ranges - no
    pack copy before the all-reduce, no copy back - and the optimizer's pointer table never changes from step to step,
    which together with the device-side dropout counter makes the whole training step HIP-graph capturable;
  * a parameter that receives several contributions in one backward (the word-embedding matrix is also the MLM
    decoder, reference vilbert.py:1463-1469) is accumulated in place by the producing kernels instead of by a torch add.

Protocol used by autograd_ops.py (``claim`` / ``result``):
    view, mode = arena.claim(param)       # None -> parameter not managed, caller uses a private buffer
      mode "fresh":  first contribution of this backward pass and ``param.grad is None``: the kernel adds into the
                     zeroed view and the Function RETURNS an alias of it (autograd installs it as ``param.grad``);
      mode "accum":  a later contribution of the same pass, or ``param.grad`` already IS the arena view (gradient
                     accumulation over micro-batches): the kernel adds into the view, the Function returns ``None``.
If ``param.grad`` is some foreign tensor (a user assigned it), the parameter is left to plain autograd.

Limits: a pass is keyed on autograd's graph-task id, so a NESTED backward (re-entrant checkpointing) inside a pass would
re-begin it; the native autograd nodes never nest and the package does not checkpoint, so this is not supported.
"""
import weakref

import torch

from . import _native as _N

# param.data_ptr() -> (weak reference to the arena, index). Weak: an arena lives exactly as long as its owner (the
# optimizer / data-parallel wrapper that built it) or a gradient view of it does; a long-lived process that builds many
# models (test suites, bench legs, sweeps) must not keep every model-sized gradient buffer alive through this table.
_BY_PTR = {}


def lookup(param):
    e = _BY_PTR.get(param.data_ptr())
    if e is None:
        return None
    arena = e[0]()
    if arena is None or arena.flat is None:
        _BY_PTR.pop(param.data_ptr(), None)
        return None
    if arena.params[e[1]] is not param:
        return None                  # the address was recycled by another tensor
    return arena, e[1]


class GradArena(object):
    def __init__(self, params, align_elems=4):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("GradArena needs at least one parameter")
        dev, dt = params[0].device, params[0].dtype
        if dt != torch.float32:
            raise RuntimeError("GradArena holds fp32 gradients")
        self._cuda = dev.type == "cuda"     # (CPU tensors only in the gloo tests of the data-parallel wrapper)
        self.align_elems = int(align_elems)
        self.params, self.offsets, self.shapes, seen, total = [], [], [], set(), 0
        for p in params:
            if id(p) in seen:
                continue
            seen.add(id(p))
            if p.device != dev or p.dtype != dt:
                raise RuntimeError("GradArena: all parameters must live on one device in fp32")
            self.params.append(p)
            self.offsets.append(total)
            self.shapes.append(p.shape)
            total += (p.numel() + align_elems - 1) // align_elems * align_elems   # every slice stays 16-byte aligned
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        self.views = [self.flat[o:o + sh.numel()].view(sh) for sh, o in zip(self.shapes, self.offsets)]
        self._written = set()          # indices claimed in the running backward pass
        self._pass_id = None           # autograd graph-task id of the running pass
        self._clean = True             # flat is all zeros
        self._zero_stream = None
        self._zero_event = None
        self._fill_seen = set()        # raw streams already ordered behind this pass's zero fill
        self._dev_index = dev.index if self._cuda else None
        self._listeners = []           # objects with .arena_written(index) / .arena_backward_done()
        self._taken_over = 0
        me = weakref.ref(self)
        for i, p in enumerate(self.params):
            old = _BY_PTR.get(p.data_ptr())
            old_arena = old[0]() if old is not None else None
            if old_arena is not None and old_arena is not self and old_arena.flat is not None and old_arena.params[old[1]] is p:
                # a newer arena (e.g. the data-parallel wrapper's bucket layout) takes the parameter over; an arena
                # that lost all its parameters frees its buffer
                old_arena._taken_over += 1
                if old_arena._taken_over >= len(old_arena.params):
                    old_arena.flat, old_arena.views = None, []
            _BY_PTR[p.data_ptr()] = (me, i)

    # ------------------------------------------------------------------------------------------------------
    def add_listener(self, obj):
        self._listeners.append(obj)

    def release(self):
        """Stop managing the parameters (their current .grad tensors stay valid views of the buffer)."""
        for p in self.params:
            e = _BY_PTR.get(p.data_ptr())
            if e is not None and e[0]() in (self, None):
                del _BY_PTR[p.data_ptr()]

    def __del__(self):
        try:
            self.release()
        except Exception:      # interpreter shutdown
            pass

    def owns(self, param, index):
        g = param.grad
        return g is not None and g.data_ptr() == self.views[index].data_ptr()

    def enter_pass(self):
        """Make sure the running backward pass has been begun (arena zero-filled unless accumulating). Called by claim()
        and by anything that writes into the arena on its own during a backward pass - the data-parallel wrapper copies
        the gradients of foreign (plain torch) autograd nodes into their slices and must not do so BEFORE the fill.
        -> False outside a backward pass."""
        task = torch._C._current_graph_task_id()
        if task == -1:
            return False
        if self._pass_id != task:
            self._pass_id = task
            self._begin_pass()
        return True

    def wait_for_fill(self):
        """Writers on another stream than the one that zero-filled the arena must wait for the fill."""
        if self._cuda and self._zero_event is not None:
            raw = _N.raw_stream(self._dev_index)
            if raw not in self._fill_seen:      # (a stream that waited once is ordered behind the fill for good)
                self._fill_seen.add(raw)
                torch.cuda.current_stream(self.flat.device).wait_event(self._zero_event)

    def note_foreign_write(self, index):
        """The caller is about to write slice `index` itself during a backward pass (see enter_pass)."""
        if self.enter_pass():
            self.wait_for_fill()
            self._written.add(index)

    def _begin_pass(self):
        """First claim of a backward pass: zero the arena unless gradients are being accumulated into it."""
        self._written.clear()
        accumulating = any(self.owns(p, i) for i, p in enumerate(self.params))
        # an event of an earlier pass (possibly recorded inside a graph capture) must never order this pass
        self._zero_event, self._zero_stream = None, None
        self._fill_seen = set()
        if not accumulating:
            if not self._clean:
                self.flat.zero_()
            if self._cuda:
                self._zero_stream = torch.cuda.current_stream(self.flat.device)
                self._zero_event = torch.cuda.Event()
                self._zero_event.record(self._zero_stream)
                self._fill_seen.add(self._zero_stream.cuda_stream)
        self._clean = False
        self._accumulating = accumulating
        torch.autograd.Variable._execution_engine.queue_callback(self._end_pass)

    def _end_pass(self):
        self._pass_id = None
        self._written.clear()
        for hook in END_PASS_HOOKS:      # e.g. join the weight-gradient side streams (autograd_ops.py)
            hook()
        for l in self._listeners:
            l.arena_backward_done()

    def claim(self, param):
        """-> (view, mode) with mode in {"fresh", "accum"}, or (None, None) if the caller should use its own buffer."""
        e = lookup(param)
        if e is None or e[0] is not self:
            return None, None
        return self._claim_at(param, e[1])

    def _claim_at(self, param, i):
        if not self.enter_pass():
            return None, None                      # not inside a backward pass (manual call of a backward op)
        self.wait_for_fill()
        if i in self._written:
            return self.views[i], "accum"
        g = param.grad
        if g is None:
            if self._accumulating:
                self.views[i].zero_()              # a stale slice next to slices that are being accumulated
            self._written.add(i)
            return self.views[i], "fresh"
        if g.data_ptr() == self.views[i].data_ptr():
            self._written.add(i)
            return self.views[i], "accum"
        return None, None

    def alias(self, index):
        """A fresh tensor object over the slice (autograd steals it: use_count 1, no copy)."""
        return self.views[index].detach()


# callables run at the end of every backward pass that wrote into an arena (engine callback, on the thread and stream that
# called backward), before the arena's listeners
END_PASS_HOOKS = []


def claim(param):
    """(view, mode, arena, index) for `param`, or (None, None, None, None)."""
    e = lookup(param)
    if e is None:
        return None, None, None, None
    view, mode = e[0]._claim_at(param, e[1])
    if view is None:
        return None, None, None, None
    return view, mode, e[0], e[1]


def claim_many(params):
    """claim() for every parameter of a list, with the per-pass work (enter_pass, wait_for_fill) done once per arena instead of
    once per parameter (round 6: a whole-layer backward node claims 16 - 22 slices at once; 509 claims per step were 1.4 ms of
    the host's 15 ms at the per-GPU batch 64). Same results as [claim(p) for p in params]."""
    out = []
    ready = None                                       # the arena whose pass has been entered by this call
    for param in params:
        e = _BY_PTR.get(param.data_ptr())
        arena = e[0]() if e is not None else None
        if arena is None or arena.flat is None or arena.params[e[1]] is not param:
            out.append(claim(param))                   # (unknown / stale entry: the careful path)
            continue
        i = e[1]
        if arena is not ready:
            if not arena.enter_pass():
                out.append((None, None, None, None))
                continue
            arena.wait_for_fill()
            ready = arena
        if i in arena._written:
            out.append((arena.views[i], "accum", arena, i))
            continue
        g = param.grad
        if g is None:
            if arena._accumulating:
                arena.views[i].zero_()
            arena._written.add(i)
            out.append((arena.views[i], "fresh", arena, i))
        elif g.data_ptr() == arena.views[i].data_ptr():
            arena._written.add(i)
            out.append((arena.views[i], "accum", arena, i))
        else:
            out.append((None, None, None, None))
    return out


def result(mode, arena, index, fallback):
    """What a backward node returns for a parameter whose gradient it wrote through claim()."""
    if mode == "fresh":
        return arena.alias(index)
    if mode == "accum":
        return None
    return fallback
