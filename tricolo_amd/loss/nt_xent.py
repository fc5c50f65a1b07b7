"""NTXentLoss on the fused gfx950 kernel - drop-in for /root/reference/tricolo/loss/nt_xent.py:10-74.

Same constructor (temperature, alpha_weight) and forward(zis, zjs, norm=True) -> 0-d tensor with autograd.  The first
argument is the `alpha_weight` side (row cross-entropy), exactly as the reference (nt_xent.py:71-74).
"""
import torch

from .. import ops
from ..layers import TriModule, require_gpu


class _NTXentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, zis, zjs, temperature, alpha, norm):
        zis, zjs = zis.contiguous(), zjs.contiguous()
        loss, ws = ops.ntxent_fwd(zis, zjs, temperature, alpha, norm)
        ctx.save_for_backward(zis, zjs, ws)          # S, the log-sum-exps and the normalised rows stay in the workspace
        ctx.hyper = (temperature, alpha, norm)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        zis, zjs, ws = ctx.saved_tensors
        t, a, n = ctx.hyper
        # one launch; the upstream scalar is folded into the kernel's coefficients (no elementwise grad * dloss passes)
        dza, dzb = ops.ntxent_bwd(zis, zjs, ws, t, a, n, dloss=dloss.contiguous().to(torch.float32))
        return dza, dzb, None, None, None


class NTXentLoss(TriModule):
    def __init__(self, temperature, alpha_weight):
        super().__init__()
        self.temperature = temperature
        self.alpha_weight = alpha_weight

    def forward(self, zis, zjs, norm=True):
        require_gpu(zis, "NTXentLoss")
        if zis.shape != zjs.shape:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(zis.shape)} vs {tuple(zjs.shape)})")
        return _NTXentFn.apply(zis, zjs, float(self.temperature), float(self.alpha_weight), bool(norm))
