"""Fused Adam on the gfx950 kernel: torch.optim.Adam semantics (L2 weight decay folded into the gradient, bias
correction, eps added after the sqrt) as instantiated by /root/reference/config/config.yaml:50-53 through
tricolo_net.py:43-44.  The step counter lives on the device, so the whole optimizer step is HIP-graph capturable.
Select it with ``optimizer._target_: tricolo_amd.optim.FusedAdam`` (the default of tricolo_amd/config/config.yaml);
``torch.optim.Adam`` keeps working on the same parameters.
"""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step_dev = None

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = closure() if closure is not None else None
        self.prepare()
        if self._step_dev is None:
            return loss
        ops.adam_tick(self._step_dev)                       # step += 1 on the device, once per optimizer step
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adam_step(p, g, st["exp_avg"], st["exp_avg_sq"], self._step_dev, group["lr"], b1, b2, group["eps"],
                              group["weight_decay"], grad_scale)
        return loss

    def prepare(self):
        """Allocate state ahead of HIP-graph capture (capture must not see first-use allocations of the counter)."""
        for group in self.param_groups:
            for p in group["params"]:
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam: parameter is not on a GPU (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if self._step_dev is None:
                    self._step_dev = torch.zeros((1,), dtype=torch.int32, device=p.device)
