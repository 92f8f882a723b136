"""Whole training step as ONE HIP graph launch.

At the reference's own per-GPU batch (global 512 over 8 GPUs = 64 samples, reference train_concap.py:290-294) the
eager step is bound by the host: ~2,600 kernel launches + autograd bookkeeping per step against ~30 ms of GPU work.
Everything in the native layer only enqueues on the current stream and never allocates or synchronises behind torch's
back, so forward + backward + optimizer can be captured once and replayed with a single ``hipGraphLaunch``:

  * the labelled rows are gathered into fixed-capacity buffers (``model.label_capacity``, torch.nonzero_static) - no
    host sync, static shapes;
  * the gradients live at fixed addresses (arena.py) and ``param.grad`` never changes identity, so the optimizer's
    pointer table is static; only its per-step scalars (learning-rate schedule, bias correction) are rewritten by the
    host into the pinned buffer a captured copy node reads (AdamW.prepare_replay);
  * the dropout masks are functions of (seed, element index): the host seeds are frozen into the graph, the graph's first
    node increments a DEVICE step counter that every dropout kernel mixes into its seed (vb_set_seed_epoch), so each
    replay draws fresh masks and forward / backward of a step still agree.

Usage (same arguments as ``model(*inputs)``; ``loss_fn`` maps the model's outputs to the scalar that is
back-propagated, default = sum of the means of the three pre-training losses as in train_concap.py:555-566)::

    step = GraphedTrainStep(model, optimizer, example_inputs)
    for batch in loader:
        loss = step(*batch)          # copies the batch into the static inputs, replays the graph
"""
import weakref

import torch

from . import _native as N


def _default_loss(outputs):
    return sum(o.mean() for o in outputs[:3])


_ACTIVE = {"epoch_ptr": None}      # the device counter currently registered with vb_set_seed_epoch by this module


def _release(base, prev_capacity, epoch_ptr):
    """Finaliser of a GraphedTrainStep (close(), garbage collection or the end of a with-block): the process-global
    dropout step counter must not keep pointing at this step's device memory (every later dropout launch of the process
    would read a freed / recycled address, and forward / backward masks of one step could then disagree), and the model
    goes back to the exact (host-synchronising) label gather."""
    if _ACTIVE["epoch_ptr"] == epoch_ptr:      # (a newer step may have registered its own counter since)
        N.lib().vb_set_seed_epoch(None)
        _ACTIVE["epoch_ptr"] = None
    if base is not None and hasattr(base, "label_capacity"):
        base.label_capacity = prev_capacity


class GraphedTrainStep(object):
    """``with GraphedTrainStep(model, opt, batch) as step: ...`` or ``step.close()`` when done; a step that is simply
    dropped is cleaned up by its finaliser."""

    def __init__(self, model, optimizer, example_inputs, loss_fn=None, label_capacity=0.25, warmup=3, check_every=1,
                 branches=None):
        """branches: "chain" = the step is captured as ONE chain of kernel nodes - the text || image fork of the encoder runs
        sequentially inside the capture; "fork" = the fork becomes parallel branches of the graph; None (default) = "chain" in
        the bf16 training mode, "fork" otherwise. Measured at batch 64 (profiles/r05_b64_graph_chain.txt): the chain replays in
        18.9 ms (bf16 mode; eager 23.7 ms, host-bound; forked graph 26.7 ms) - this HIP runtime serialises cross-branch edges at
        replay at a cost that exceeds what the overlap buys (the effect GraphedForward(branches="auto") measures per instance).
        fp32: chain 37.3 / fork 42.0 / eager 32.0 ms - GPU-bound, a graph does not pay there. (Round 5's wrong losses of the fp8
        chain after memory had been freed before the capture were MEMSET NODES that this runtime does not execute at replay in
        that situation - tools/memset_node_repro.py; every zero fill on a captured path is a kernel since round 6, DESIGN.md
        section 4.5, tests/test_graphed_gpu.py::test_chain_graph_after_freed_memory...)"""
        base = model.module if hasattr(model, "module") else model
        bf16 = N.bf16_stream() or bool(getattr(base, "_vb_bf16", False))      # (process-wide mode, or model.half())
        if branches is None:
            branches = "chain" if bf16 else "fork"
        if branches not in ("chain", "fork"):
            raise ValueError("branches: chain | fork")
        if bf16 and warmup < 2:
            # the device table of the one-launch shadow refresh is built by the first forward AFTER an optimizer step
            # (ops16.refresh_stale): that must be an eager warm-up step, not the capture
            raise ValueError("GraphedTrainStep in the bf16 mode needs warmup >= 2")
        self.branches = branches
        self.model, self.opt = model, optimizer
        self.loss_fn = loss_fn or _default_loss
        prev_capacity = getattr(base, "label_capacity", None)
        if hasattr(base, "label_capacity"):
            base.label_capacity = label_capacity
        self._base = base
        self.check_every = max(0, int(check_every))
        dev = example_inputs[0].device
        if dev.type != "cuda":
            raise RuntimeError("GraphedTrainStep needs HIP-device inputs - no CPU fallback")
        self.static = [t.clone() if torch.is_tensor(t) else t for t in example_inputs]
        # device step counter of the dropout masks
        self.epoch = torch.zeros(1, dtype=torch.int64, device=dev)
        N.check(N.lib().vb_set_seed_epoch(self.epoch.data_ptr()), "vb_set_seed_epoch")
        _ACTIVE["epoch_ptr"] = self.epoch.data_ptr()
        # the finaliser must not reference self (it would never run); the epoch tensor is kept alive by self only, so
        # the registration is dropped no later than the memory it points to
        self._finalizer = weakref.finalize(self, _release, base, prev_capacity, self.epoch.data_ptr())
        # overflow flag of the fixed-capacity label gather: a pinned host word the captured graph refreshes every replay
        self._overflow_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        # Warm-up and capture must not train: parameters and optimizer state are snapshotted here and restored after
        # the capture (the warm-up steps are real eager steps on the example batch; the capture itself executes nothing
        # on the device but advances the host-side step counts).
        snap_p = [p.detach().clone() for g in optimizer.param_groups for p in g["params"]]
        snap_s = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in optimizer.state[p].items()}
                  for g in optimizer.param_groups for p in g["params"] if p in optimizer.state}
        # eager warm-up on a side stream (allocator pools, optimizer state, gradient arena, GEMM plans) - the
        # documented pattern for capturing a whole network
        # (the weight-gradient side streams stay out of the graph: every cross-stream edge costs at replay - measured
        # 1,645 samples/s with them vs 1,730 without at batch 64 - and a replay has no launch gaps for them to fill)
        from . import autograd_ops as _A
        from . import vilbert as _V
        prev_ws = _A.set_wgrad_stream(False)
        prev_fork = _V._TWO_STREAMS_IN_GRAPH
        _V._TWO_STREAMS_IN_GRAPH = prev_fork and branches == "fork"
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._eager_step()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            optimizer.zero_grad(set_to_none=True)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = self._eager_step()
            # the captured split-K launches store into the deterministic workspace of this device (address and slice baked
            # into the graph): hold it for as long as the graph can be replayed (_native also retires, never frees, them)
            self._det_workspace = N.deterministic_workspace(dev)
        finally:
            _A.set_wgrad_stream(prev_ws)
            _V._TWO_STREAMS_IN_GRAPH = prev_fork
        with torch.no_grad():
            i = 0
            for g in optimizer.param_groups:
                for p in g["params"]:
                    p.copy_(snap_p[i])
                    i += 1
                    st = optimizer.state.get(p)
                    if st is None:
                        continue
                    old = snap_s.get(id(p))
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            v.copy_(old[k]) if old is not None else v.zero_()      # same tensors: the graph holds their addresses
                        else:
                            st[k] = old[k] if old is not None else 0
            self.epoch.zero_()
        del snap_p, snap_s
        self.replays = 0

    def _eager_step(self):
        N.check(N.lib().vb_bump_counter(N.stream_ptr(), self.epoch.data_ptr()), "vb_bump_counter")
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model(*self.static))
        flag = getattr(self._base, "_label_overflow", None)
        if flag is not None:
            self._overflow_host.copy_(flag, non_blocking=True)   # (a D2H copy node inside the captured graph)
        loss.backward()
        self.opt.step()
        return loss

    def __call__(self, *inputs):
        if len(inputs) != len(self.static):
            raise RuntimeError("GraphedTrainStep: expected %d inputs" % len(self.static))
        # the previous replay must be done with the optimizer's pinned table before the host rewrites it
        torch.cuda.current_stream().synchronize()
        # ... which also makes the previous step's overflow word valid: dropped label rows would bias the losses, so
        # this is checked by default (check_every=1 costs one host read, no device work); 0 = never
        if self.check_every and self.replays % self.check_every == 0 and int(self._overflow_host[0]) != 0:
            raise RuntimeError("GraphedTrainStep: the labelled rows of the previous batch exceeded the fixed gather "
                               "capacity (label_capacity=%s of the positions) - rows were dropped; construct the step "
                               "with a larger label_capacity" % (self._base.label_capacity,))
        for s, t in zip(self.static, inputs):
            if torch.is_tensor(s) and s.data_ptr() != t.data_ptr():
                if s.shape != t.shape or s.dtype != t.dtype:
                    raise RuntimeError("GraphedTrainStep: input shape / dtype differs from the captured one")
                s.copy_(t, non_blocking=True)
        self.opt.prepare_replay()
        self.graph.replay()
        N.weights_changed()          # the replayed optimizer kernel rewrote the parameters (caches of derived weights)
        self.replays += 1
        return self.loss

    def check(self):
        """Off the hot path: raises if the fixed-capacity label gather of the last step overflowed."""
        if hasattr(self._base, "check_label_capacity"):
            self._base.check_label_capacity()

    def close(self):
        self._finalizer()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class GraphedForward(object):
    """Inference forward (eval, no_grad) as ONE HIP graph launch: at the per-GPU batch of BASELINE configs[4]
    (1024 / 8 = 128 samples, fp8 linears) the ~450 launches of a forward take ~6.6 ms to issue against ~4 ms of GPU work.
    ``fwd = GraphedForward(model, example_inputs); out = fwd(*batch)`` - the outputs are static tensors, overwritten by
    the next call. In fp8 mode the quantised weights are taken from the cache that the warm-up filled, so the graph
    contains no weight quantisation (call again after the weights change).

    branches: "fork" = the text || image fork of the encoder becomes two parallel branches of the graph, "chain" = one
    chain of nodes, "auto" (default) = the chain is captured and timed here (a few replays), then the fork up to
    `fork_attempts` times until one instance beats the chain; the fastest instance is kept.
    Why instances of the SAME fork graph differ (tools/mx_graph_bisect.py, profiles/r05_mx_graph_bisect.txt; this was the
    "12.6 k vs 34 k samples/s" puzzle of the round-4 bench line): the replay rate of a forked graph is bimodal PER
    INSTANTIATED GRAPH - at batch 128 in the MX mode 12.5 k or 33.9 k samples/s, stable over all replays of one instance,
    while the chain always replays at 30.0 k. Which mode an instance gets does not depend on what ran before (extra
    streams, RCCL, training steps, other graphs - all tried) but flips with the number of graph instantiations / stream
    creations since the last one: the runtime binds the branch of a graph to one of its hardware queues round-robin when
    the graph is instantiated, and when the side branch lands on the queue of the launching stream the two branches'
    persistent kernels (one block per CU each) alternate instead of overlapping and every fork / join edge becomes a
    cross-queue barrier. Capturing again advances the assignment, so a second attempt gets the fast binding.
    `self.branches` says which form was kept, `self.trial_ms` what every attempt measured."""

    def __init__(self, model, example_inputs, warmup=2, branches="auto", trial_replays=6, fork_attempts=3):
        from . import vilbert as _V
        dev = example_inputs[0].device
        if dev.type != "cuda":
            raise RuntimeError("GraphedForward needs HIP-device inputs - no CPU fallback")
        if model.training:
            raise RuntimeError("GraphedForward captures an inference forward: call model.eval() first")
        if branches not in ("auto", "fork", "chain"):
            raise ValueError("branches: auto | fork | chain")
        self.model = model
        self.static = [t.clone() if torch.is_tensor(t) else t for t in example_inputs]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                model(*self.static)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)

        def capture(fork):
            prev = _V._TWO_STREAMS_IN_GRAPH
            _V._TWO_STREAMS_IN_GRAPH = bool(fork)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.no_grad(), torch.cuda.graph(g):
                    out = model(*self.static)
            finally:
                _V._TWO_STREAMS_IN_GRAPH = prev
            return g, out

        def trial(g):
            g.replay()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(trial_replays):
                g.replay()
            e1.record()
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1) / trial_replays

        self.trial_ms = {}
        forks = _V._TWO_STREAMS and _V._TWO_STREAMS_IN_GRAPH       # (with the overlap switched off there is only the chain)
        if branches == "chain" or not forks:
            self.branches = "chain"
            self.graph, self.out = capture(False)
            return
        best = None                                   # (form, graph, outputs, ms per replay)
        if branches == "auto":
            g, out = capture(False)
            best = ("chain", g, out, trial(g))
            self.trial_ms["chain"] = best[3]
        # forced "fork": two instances are compared with each other (the slow binding is 2.7x slower - unmistakable)
        attempts = max(1, fork_attempts) if branches == "auto" else max(1, min(2, fork_attempts))
        for attempt in range(attempts):
            g, out = capture(True)
            ms = trial(g)
            self.trial_ms["fork#%d" % attempt] = ms
            if best is None or ms < best[3]:
                best = ("fork", g, out, ms)
            del g, out
            if branches == "auto" and best[0] == "fork":
                break                                 # this instance beats the chain: the fast binding
        self.branches, self.graph, self.out = best[:3]

    def __call__(self, *inputs):
        if len(inputs) != len(self.static):
            raise RuntimeError("GraphedForward: expected %d inputs" % len(self.static))
        for s, t in zip(self.static, inputs):
            if torch.is_tensor(s) and s.data_ptr() != t.data_ptr():
                if s.shape != t.shape or s.dtype != t.dtype:
                    raise RuntimeError("GraphedForward: input shape / dtype differs from the captured one")
                s.copy_(t, non_blocking=True)
        self.graph.replay()
        return self.out
