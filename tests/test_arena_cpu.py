"""Gradient arena protocol (vilbert/arena.py) on CPU tensors with a toy backward node that behaves like the native
ones (adds its weight gradient into the claimed slice): zero copy into param.grad, one fill per backward pass, tied
parameters accumulated in place, gradient accumulation over micro-batches, foreign gradients left alone; and the
data-parallel wrapper's reaction to a gradient that arrives after its bucket was reduced."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
from torch import nn
from torch.autograd import Function

from vilbert import arena as A


class ToyLinear(Function):
    """y = x @ w.T with the weight gradient ADDED into its arena slice (like the split-K wgrad kernel)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        view, mode, ar, idx = A.claim(w)
        dw = dy.t() @ x
        if view is None:
            return dy @ w, dw
        view.add_(dw)                                   # the native kernels add into the (zeroed) slice
        return dy @ w, A.result(mode, ar, idx, None)


def test_arena_zero_copy_tied_parameters_and_accumulation():
    torch.manual_seed(0)
    w1, w2 = nn.Parameter(torch.randn(4, 3)), nn.Parameter(torch.randn(3, 4))
    ar = A.GradArena([w2, w1])
    x = torch.randn(5, 3)

    def loss():   # w1 is used twice (a tied parameter: two contributions per backward)
        h = ToyLinear.apply(x, w1)
        return (ToyLinear.apply(ToyLinear.apply(h, w2), w1) ** 2).sum()

    def reference():
        a, b = w1.detach().clone().requires_grad_(True), w2.detach().clone().requires_grad_(True)
        (((x @ a.t()) @ b.t() @ a.t()) ** 2).sum().backward()
        return a.grad, b.grad

    r1, r2 = reference()
    loss().backward()
    i1, i2 = A.lookup(w1)[1], A.lookup(w2)[1]
    assert w1.grad.data_ptr() == ar.views[i1].data_ptr() and w2.grad.data_ptr() == ar.views[i2].data_ptr()   # no copy
    assert torch.allclose(w1.grad, r1, atol=1e-5) and torch.allclose(w2.grad, r2, atol=1e-5)
    # a second backward without zeroing accumulates (micro-batches)
    loss().backward()
    assert torch.allclose(w1.grad, 2 * r1, atol=1e-4) and torch.allclose(w2.grad, 2 * r2, atol=1e-4)
    # set_to_none + backward: the arena is refilled with zeros once, the gradients are fresh again
    w1.grad = w2.grad = None
    loss().backward()
    assert torch.allclose(w1.grad, r1, atol=1e-5) and torch.allclose(w2.grad, r2, atol=1e-5)
    # only one of the two zeroed: the stale slice is cleared, the other keeps accumulating
    w1.grad = None
    loss().backward()
    assert torch.allclose(w1.grad, r1, atol=1e-5) and torch.allclose(w2.grad, 2 * r2, atol=1e-4)
    # a foreign gradient tensor: the arena steps aside and autograd accumulates as usual
    w1.grad, w2.grad = torch.ones(4, 3), None
    loss().backward()
    assert torch.allclose(w1.grad, 1 + r1, atol=1e-5) and w1.grad.data_ptr() != ar.views[i1].data_ptr()
    ar.release()
    assert A.lookup(w1) is None


def test_newer_arena_takes_parameters_over():
    w = nn.Parameter(torch.randn(2, 2))
    a1 = A.GradArena([w])
    a2 = A.GradArena([w])
    assert A.lookup(w)[0] is a2 and a1.flat is None
    a2.release()


class Heads(nn.Module):
    def __init__(self):
        super().__init__()
        self.body, self.head_a, self.head_b = nn.Linear(4, 4), nn.Linear(4, 2), nn.Linear(4, 2)

    def forward(self, x, which):
        h = torch.relu(self.body(x))
        return (self.head_a(h) if which == 0 else self.head_b(h)).pow(2).mean()


def test_ddp_gradient_after_bucket_launch_is_loud():
    """The used-parameter set is tracked per bucket as a SET: a parameter that starts receiving gradients after its
    bucket was reduced early must raise instead of being silently replaced by stale data."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from vilbert.distributed import DistributedDataParallel as DDP
        torch.manual_seed(1)
        net = Heads()
        ddp = DDP(net, message_size=1)            # one bucket per parameter
        x = torch.randn(3, 4)
        for _ in range(2):                         # learns: head_b is never used
            ddp.zero_grad()
            ddp(x, 0).backward()
            assert net.head_b.weight.grad is None and net.head_a.weight.grad is not None
        ref = Heads()
        ref.load_state_dict(net.state_dict())
        ref(x, 0).backward()
        assert torch.allclose(net.body.weight.grad, ref.body.weight.grad, atol=1e-6)
        # switching the head: head_b's bucket is not expected to receive anything -> it is only reduced at the end
        # of backward (fine), but body's bucket IS reduced early while head_a (expected) never arrives: allowed, the
        # learnt set is refreshed; the gradients must still be right
        ddp.zero_grad()
        ddp(x, 1).backward()
        ref.zero_grad()
        ref(x, 1).backward()
        assert torch.allclose(net.head_b.weight.grad, ref.head_b.weight.grad, atol=1e-6)
        assert torch.allclose(net.body.weight.grad, ref.body.weight.grad, atol=1e-6)
        # delayed mode never launches early
        ddp2 = DDP(Heads(), delay_allreduce=True, message_size=1)
        for which in (0, 1, 0):
            ddp2.zero_grad()
            ddp2(x, which).backward()
        # one bucket holding everything: after learning {body, head_a} the bucket is reduced as soon as those arrive;
        # a later gradient for head_b (same bucket) must raise
        net3 = Heads()
        ddp3 = DDP(net3, message_size=10 ** 9)

        class Late(nn.Module):   # forces head_b's gradient to arrive LAST (it is applied to the network input)
            def forward(self, x, which):
                if which == 0:
                    return net3(x, 0)
                return net3(torch.relu(net3.head_b(x)) @ torch.ones(2, 4), 0)
        late = Late()
        for _ in range(2):
            ddp3.zero_grad()
            late(x, 0).backward()
        ddp3.zero_grad()
        with pytest.raises(RuntimeError, match="delay_allreduce"):
            late(x, 1).backward()
    finally:
        dist.destroy_process_group()


def test_ddp_foreign_head_gradients_survive_the_arena_fill():
    """Round-2 advisor (high): a plain torch head on top of a native body hands its gradient to the wrapper BEFORE the
    first native claim() of the pass; from the second step on (zero_grad(set_to_none=True) -> no slice is owned) the
    arena fill of that first claim used to wipe the copied slice, and the head trained on zeros. The wrapper now begins
    the arena's pass before it copies. Three steps, head and body gradients against plain autograd."""

    class Body(nn.Module):
        def __init__(self):
            super().__init__()
            self.w = nn.Parameter(torch.randn(5, 4))

        def forward(self, x):
            return ToyLinear.apply(x, self.w)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.body, self.head = Body(), nn.Linear(5, 2)       # head = foreign (plain torch) autograd nodes

        def forward(self, x):
            return self.head(torch.tanh(self.body(x))).pow(2).mean()

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from vilbert.distributed import DistributedDataParallel as DDP
        torch.manual_seed(3)
        net = Net()
        ref = Net()
        ref.load_state_dict(net.state_dict())
        ddp = DDP(net, message_size=10 ** 9)
        x = torch.randn(6, 4)
        for step in range(3):
            ddp.zero_grad()                       # set_to_none=True
            ddp(x).backward()
            ref.zero_grad()
            h = torch.tanh(x @ ref.body.w.t())
            ref.head(h).pow(2).mean().backward()
            for (n, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
                assert p.grad is not None and torch.allclose(p.grad, r.grad, atol=1e-6), (step, n, p.grad, r.grad)
                assert p.grad.abs().max() > 0, (step, n)
            with torch.no_grad():
                for p, r in zip(net.parameters(), ref.parameters()):
                    p -= 0.1 * p.grad
                    r -= 0.1 * r.grad
        ddp.arena.release()
    finally:
        dist.destroy_process_group()


def test_registry_does_not_keep_arenas_alive():
    """Round-2 advisor (low): the pointer registry holds arenas weakly - dropping the owner frees the gradient buffer."""
    import gc
    import weakref
    w = nn.Parameter(torch.randn(3, 3))
    ar = A.GradArena([w])
    ref = weakref.ref(ar)
    assert A.lookup(w)[0] is ar
    del ar
    gc.collect()
    assert ref() is None and A.lookup(w) is None


class ToyBlock(Function):
    """Two linears as ONE node that claims all its targets at once (claim_many), like the whole-layer nodes of layers.py."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3):
        ctx.save_for_backward(x, w1, w2, w3)
        return (x @ w1.t()) @ w2.t() + 0.0 * w3.sum()

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, w3 = ctx.saved_tensors
        h = x @ w1.t()
        claims = A.claim_many([w1, w2, w3])
        grads = [(dy @ w2).t() @ x, dy.t() @ h, torch.zeros_like(w3)]
        out = []
        for (view, mode, ar, idx), g in zip(claims, grads):
            if view is None:
                out.append(g)
            else:
                view.add_(g)
                out.append(A.result(mode, ar, idx, None))
        return ((dy @ w2) @ w1,) + tuple(out)


def test_claim_many_is_claim_for_every_parameter():
    """arena.claim_many (round 6: one pass-entry per arena for the 16 - 22 targets of a whole-layer backward node) must behave
    exactly like claim() per parameter: fresh / accumulating / tied / foreign gradients, a parameter outside the arena, and
    outside a backward pass."""
    torch.manual_seed(1)
    w1, w2, w3 = nn.Parameter(torch.randn(4, 3)), nn.Parameter(torch.randn(2, 4)), nn.Parameter(torch.randn(5))
    ar = A.GradArena([w1, w2])                         # w3 is NOT managed
    x = torch.randn(6, 3)
    assert A.claim_many([w1, w2, w3]) == [(None, None, None, None)] * 3       # not inside a backward pass

    def loss():
        return (ToyBlock.apply(x, w1, w2, w3) ** 2).sum() + (ToyLinear.apply(x, w1) ** 2).sum()     # w1: a second, per-op writer

    def reference():
        a, b = w1.detach().clone().requires_grad_(True), w2.detach().clone().requires_grad_(True)
        ((((x @ a.t()) @ b.t()) ** 2).sum() + ((x @ a.t()) ** 2).sum()).backward()
        return a.grad, b.grad
    r1, r2 = reference()
    loss().backward()
    i1, i2 = A.lookup(w1)[1], A.lookup(w2)[1]
    assert w1.grad.data_ptr() == ar.views[i1].data_ptr() and w2.grad.data_ptr() == ar.views[i2].data_ptr()
    assert torch.allclose(w1.grad, r1, atol=1e-5) and torch.allclose(w2.grad, r2, atol=1e-5)
    assert w3.grad is not None and float(w3.grad.abs().max()) == 0.0 and A.lookup(w3) is None
    loss().backward()                                   # accumulation into owned slices
    assert torch.allclose(w1.grad, 2 * r1, atol=1e-4) and torch.allclose(w2.grad, 2 * r2, atol=1e-4)
    w1.grad, w2.grad, w3.grad = None, torch.ones(2, 4), None          # one fresh, one foreign
    loss().backward()
    assert torch.allclose(w1.grad, r1, atol=1e-5) and w1.grad.data_ptr() == ar.views[i1].data_ptr()
    assert torch.allclose(w2.grad, 1 + r2, atol=1e-5) and w2.grad.data_ptr() != ar.views[i2].data_ptr()
    ar.release()
