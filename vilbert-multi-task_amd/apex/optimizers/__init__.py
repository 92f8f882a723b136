"""Import-name shim for the reference's reduced-precision branch (train_concap.py:443-461, train_tasks.py the same block):

    from apex.optimizers import FP16_Optimizer, FusedAdam
    optimizer = FusedAdam(grouped_parameters, lr=..., bias_correction=False, max_grad_norm=1.0)
    optimizer = FP16_Optimizer(optimizer, dynamic_loss_scale=True)      # or static_loss_scale=...
    ...
    model.half()                    # train_concap.py:504-505
    optimizer.backward(loss)        # :570-571
    optimizer.step(); optimizer.zero_grad()

NVIDIA apex does not exist on ROCm images. On this package that mode is the bf16 stream: bfloat16
activations / gradients, fp32 master weights that the native AdamW updates directly - bf16 has fp32's exponent range, so
there is no loss scale to manage: `backward(loss)` is `loss.backward()`, `loss_scale` reads 1.0 and never overflows.
`max_grad_norm` (FusedAdam's global-norm clipping, which apex applies through FP16_Optimizer's combined scale) is one norm
over the flat gradient arena + one in-place scale of it (two torch calls on one buffer; declared in DESIGN.md section 1).
"""
import torch

from vilbert.optim import AdamW


class FusedAdam(AdamW):
    """apex.optimizers.FusedAdam's constructor on the native multi-tensor AdamW (csrc/optimizer.hip). apex's update
    p -= step_size * (m / (sqrt(v) + eps) + weight_decay * p) is AdamW's decoupled form with the decay scaled by the same
    step size; eps_inside_sqrt is not offered by the native kernel."""

    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-8, eps_inside_sqrt=False,
                 weight_decay=0.0, max_grad_norm=0.0, amsgrad=False):
        if amsgrad:
            raise RuntimeError("FusedAdam does not support the AMSGrad variant.")
        if eps_inside_sqrt:
            raise RuntimeError("FusedAdam (MI355X-native): eps_inside_sqrt is not supported")
        super(FusedAdam, self).__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                        correct_bias=bool(bias_correction))
        self.max_grad_norm = float(max_grad_norm)

    def clip_(self):
        """Global-norm clipping of every gradient this optimizer owns; returns the norm (a device scalar) or None."""
        if self.max_grad_norm <= 0.0:
            return None
        if self._arena is not None:
            flat = self._arena.flat
            norm = torch.linalg.vector_norm(flat)
            flat.mul_(torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0))
            return norm
        grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
        return torch.nn.utils.clip_grad_norm_([p for g in self.param_groups for p in g["params"] if p.grad is not None],
                                              self.max_grad_norm) if grads else None

    def step(self, closure=None, **_apex_kwargs):      # (apex passes grads / output_params / scale / grad_norms)
        self.clip_()
        return super(FusedAdam, self).step(closure)


class FP16_Optimizer(torch.optim.Optimizer):
    """apex.optimizers.FP16_Optimizer's public face around a FusedAdam (or any optimizer of this package). The wrapped
    optimizer already holds the fp32 master weights (the model's own parameters: `model.half()` leaves them fp32 here), so
    there are no fp16 copies to keep in sync and no scaled gradients to unscale. A torch Optimizer by type (the scripts hand
    it to `WarmupLinearSchedule`, whose base class insists on one): it SHARES the wrapped optimizer's parameter-group
    dictionaries and state."""

    def __init__(self, init_optimizer, static_loss_scale=1.0, dynamic_loss_scale=False, dynamic_loss_args=None, verbose=True):
        super(FP16_Optimizer, self).__init__(init_optimizer.param_groups, init_optimizer.defaults)
        assert all(a is b for a, b in zip(self.param_groups, init_optimizer.param_groups))     # the same dict objects
        self.optimizer = init_optimizer
        self.state = init_optimizer.state
        self.dynamic_loss_scale = bool(dynamic_loss_scale)
        self.static_loss_scale = static_loss_scale
        self.overflow = False
        self.cur_scale = 1.0

    @property
    def loss_scale(self):
        return self.cur_scale

    def zero_grad(self, set_grads_to_None=True):
        self.optimizer.zero_grad()

    def backward(self, loss):
        loss.backward()

    def step(self, closure=None):
        return self.optimizer.step(closure)

    def state_dict(self):
        return {"dynamic_loss_scale": self.dynamic_loss_scale, "cur_scale": self.cur_scale,
                "optimizer_state_dict": self.optimizer.state_dict()}

    def load_state_dict(self, state_dict):
        self.optimizer.load_state_dict(state_dict["optimizer_state_dict"])
