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


class _NTXentAllFn(torch.autograd.Function):
    """All pairs of M = 2 / 3 embeddings: outputs (loss_pair_0, ..., loss_pair_{P-1}, total)."""

    @staticmethod
    def forward(ctx, temperature, alpha, norm, *zs):
        zs = tuple(z.contiguous() for z in zs)
        losses, ws = ops.ntxent_multi_fwd(zs, temperature, alpha, norm)
        ctx.save_for_backward(ws, *zs)
        ctx.hyper = (temperature, alpha, norm)
        ctx.set_materialize_grads(False)             # unused outputs (normally every pair loss) arrive as None, not as zeros
        return tuple(losses[i] for i in range(losses.numel()))

    @staticmethod
    def backward(ctx, *grads):
        ws, *zs = ctx.saved_tensors
        t, a, n = ctx.hyper
        f = lambda g: None if g is None else g.contiguous().to(torch.float32)     # noqa: E731
        dpairs = [f(g) for g in grads[:-1]]
        dzs = ops.ntxent_multi_bwd(zs, ws, t, a, n, dpairs=dpairs if any(d is not None for d in dpairs) else None, dtotal=f(grads[-1]))
        return (None, None, None, *dzs)


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

    def all_pairs(self, embeddings, norm=True):
        """The losses of every pair of `embeddings` (a list in the reference's modality order: the earlier one of a pair is
        the `zis` / alpha side, tricolo_net.py:59-61) and their sum, from ONE autograd node: 4 launches forward, 1 backward,
        each embedding's gradient already summed over its pairs.  Returns ([pair losses in combination order], total), or
        None when the shape is outside the fused kernels' range (callers then loop over forward())."""
        zs = list(embeddings)
        if not (all(z.is_cuda and z.dtype == torch.float32 and z.dim() == 2 and z.shape == zs[0].shape for z in zs)
                and ops.ntxent_multi_supported(zs[0].shape[0], zs[0].shape[1], len(zs))):
            return None
        out = _NTXentAllFn.apply(float(self.temperature), float(self.alpha_weight), bool(norm), *zs)
        return list(out[:-1]), out[-1]
