"""AdamW + warm-up schedules with the semantics of pytorch-transformers 1.0.0 (the optimizer the reference
imports: train_concap.py:27,465-476; train_tasks.py:26-30,426-437), the update itself being ONE native
multi-tensor launch (csrc/optimizer.hip) instead of ~6 torch kernels per parameter tensor.

Also importable as ``pytorch_transformers.optimization`` (shim package next to this one) so that the
unchanged reference scripts pick it up.
"""
import ctypes
import math

import numpy as np
import torch
from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR

from . import _native as N

CHUNK_ELEMS = 64 * 1024


class AdamW(Optimizer):
    """Adam with decoupled weight decay; ``correct_bias=False`` reproduces the original BERT optimizer
    (train_tasks.py:426). State layout (``step``, ``exp_avg``, ``exp_avg_sq``) matches pytorch-transformers, so
    the ``.tar`` checkpoints the reference scripts write stay interchangeable."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        super(AdamW, self).__init__(params, defaults)
        self._plan_key, self._plan = None, None
        self._arena = None
        self._ensure_arena()

    def _ensure_arena(self):
        """Gives the optimizer's parameters a gradient arena (arena.py: gradients at fixed addresses in one flat
        buffer, zero-filled once per backward, written in place by the backward kernels) unless something else - the
        data-parallel wrapper - already manages them. Done at construction when the parameters are on the device,
        else at the first step()."""
        if self._arena is not None:
            return
        from . import arena
        params = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if not params or not all(p.is_cuda and p.dtype == torch.float32 for p in params):
            return
        if all(arena.lookup(p) is not None for p in params):
            return
        self._arena = arena.GradArena(list(reversed(params)))

    def _build_plan(self, entries, device):
        """Static part of the launch tables for this set of tensors: chunk lists on the device and a
        host-side structured array whose grad / hyper-parameter columns are refreshed every step."""
        tab = np.zeros(len(entries), dtype=np.dtype([
            ("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("numel", "<i8"),
            ("step_size", "<f4"), ("beta1", "<f4"), ("beta2", "<f4"), ("eps", "<f4"), ("decay", "<f4"),
            ("reserved", "<f4")]))
        assert tab.dtype.itemsize == ctypes.sizeof(N.AdamWTensor)
        chunk_t, chunk_o = [], []
        for i, (p, st, _g) in enumerate(entries):
            tab["param"][i], tab["exp_avg"][i], tab["exp_avg_sq"][i] = p.data_ptr(), st["exp_avg"].data_ptr(), \
                st["exp_avg_sq"].data_ptr()
            tab["numel"][i] = p.numel()
            for off in range(0, p.numel(), CHUNK_ELEMS):
                chunk_t.append(i)
                chunk_o.append(off)
        # two pinned staging copies of the table (the host may run one step ahead of the device) + events
        nbytes = tab.nbytes
        pinned = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        return dict(tab=tab, n_chunks=len(chunk_t), pinned=pinned, events=[None, None], turn=0,
                    dev_tab=torch.empty(nbytes, dtype=torch.uint8, device=device),
                    chunk_tensor=torch.tensor(chunk_t, dtype=torch.int32, device=device),
                    chunk_off=torch.tensor(chunk_o, dtype=torch.int64, device=device))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        entries = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                if not p.is_cuda:
                    raise RuntimeError("vilbert.optim.AdamW runs on HIP devices only - no CPU fallback")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("vilbert.optim.AdamW needs contiguous fp32 parameters")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p)
                    state["exp_avg_sq"] = torch.zeros_like(p)
                state["step"] += 1
                entries.append((p, state, group))
        if not entries:
            return loss
        self._ensure_arena()
        device = entries[0][0].device
        key = tuple((p.data_ptr(), st["exp_avg"].data_ptr()) for p, st, _ in entries)
        if key != self._plan_key:
            self._plan_key, self._plan = key, self._build_plan(entries, device)
        plan = self._plan
        capturing = torch.cuda.is_current_stream_capturing()
        keep = self._fill_table(plan, entries)
        # asynchronous upload through pinned memory: no host <-> device synchronisation in step()
        if capturing:
            # HIP-graph capture: the copy node reads pinned buffer 0 at every replay; prepare_replay() refreshes it
            k = 0
            self._captured = (plan, entries)
        else:
            k = plan["turn"]
            plan["turn"] = k ^ 1
            if plan["events"][k] is not None:
                plan["events"][k].synchronize()
        plan["pinned"][k].numpy()[:] = plan["tab"].view(np.uint8).reshape(-1)
        dev_tab = plan["dev_tab"]
        dev_tab.copy_(plan["pinned"][k], non_blocking=True)
        if not capturing:
            ev = torch.cuda.Event()
            ev.record()
            plan["events"][k] = ev
        N.check(N.lib().vb_adamw_step(N.stream_ptr(), plan["n_chunks"], dev_tab.data_ptr(),
                                      plan["chunk_tensor"].data_ptr(), plan["chunk_off"].data_ptr(), CHUNK_ELEMS),
                "vb_adamw_step")
        N.weights_changed()
        del keep
        return loss

    def _fill_table(self, plan, entries, pointers=True):
        """Per-step columns of the launch table: gradient pointers and the hyper-parameters of this step (vectorised:
        this runs on the host once per step, ~530 rows)."""
        tab = plan["tab"]
        keep = []
        if pointers:
            for i, (p, st, g) in enumerate(entries):
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(grad)
                tab["grad"][i] = grad.data_ptr()
        n = len(entries)
        gi = plan.get("group_index")
        if gi is None or len(gi) != n:
            ids = {id(g): k for k, g in enumerate(self.param_groups)}
            gi = plan["group_index"] = np.array([ids[id(g)] for _p, _st, g in entries], dtype=np.int64)
        groups = self.param_groups
        lr = np.array([g["lr"] for g in groups], dtype=np.float64)[gi]
        b1 = np.array([g["betas"][0] for g in groups], dtype=np.float64)[gi]
        b2 = np.array([g["betas"][1] for g in groups], dtype=np.float64)[gi]
        eps = np.array([g["eps"] for g in groups], dtype=np.float64)[gi]
        wd = np.array([g["weight_decay"] for g in groups], dtype=np.float64)[gi]
        corr = np.array([bool(g["correct_bias"]) for g in groups])[gi]
        steps = np.fromiter((st["step"] for _p, st, _g in entries), dtype=np.float64, count=n)
        step_size = np.where(corr, lr * np.sqrt(1.0 - b2 ** steps) / (1.0 - b1 ** steps), lr)
        tab["step_size"], tab["beta1"], tab["beta2"] = step_size, b1, b2
        tab["eps"], tab["decay"] = eps, lr * wd
        return keep

    def prepare_replay(self):
        """Host side of one replay of a captured step (GraphedTrainStep): advances the step counts and rewrites the
        pinned table the captured copy node reads (learning-rate schedule, bias correction). The previous replay
        must have finished reading the table (the caller synchronises)."""
        plan, entries = self._captured
        for _p, st, _g in entries:
            st["step"] += 1
        self._fill_table(plan, entries, pointers=False)     # the gradients live at fixed addresses (arena)
        plan["pinned"][0].numpy()[:] = plan["tab"].view(np.uint8).reshape(-1)


class ConstantLRSchedule(LambdaLR):
    def __init__(self, optimizer, last_epoch=-1):
        super(ConstantLRSchedule, self).__init__(optimizer, lambda _: 1.0, last_epoch=last_epoch)


class WarmupConstantSchedule(LambdaLR):
    """Linear warm-up 0 -> 1 over ``warmup_steps`` steps, then constant (train_tasks.py:437)."""

    def __init__(self, optimizer, warmup_steps, last_epoch=-1):
        self.warmup_steps = warmup_steps
        super(WarmupConstantSchedule, self).__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return float(step) / float(max(1.0, self.warmup_steps))
        return 1.0


class WarmupLinearSchedule(LambdaLR):
    """Linear warm-up 0 -> 1 over ``warmup_steps``, then linear decay to 0 at ``t_total``
    (train_concap.py:472-476, train_tasks.py:433-435)."""

    def __init__(self, optimizer, warmup_steps, t_total, last_epoch=-1):
        self.warmup_steps, self.t_total = warmup_steps, t_total
        super(WarmupLinearSchedule, self).__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return float(step) / float(max(1, self.warmup_steps))
        return max(0.0, float(self.t_total - step) / float(max(1.0, self.t_total - self.warmup_steps)))
