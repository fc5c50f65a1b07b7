"""Host-side building blocks shared by the encoder modules: module base class, dense layers on the MFMA conv
kernel, and helpers for the hand-written backward passes."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops

try:                                    # the reference's modules are pl.LightningModule (SURVEY 8b "Type / ownership")
    import lightning.pytorch as _pl     # not installed on the build / GPU boxes; used when present
    _Base = _pl.LightningModule
except Exception:                       # noqa: BLE001
    _Base = nn.Module


class TriModule(_Base):
    """nn.Module with the ``.device`` property the reference code relies on (bigru.py:16, nt_xent.py:62)."""

    if _Base is nn.Module:
        @property
        def device(self):
            for p in self.parameters():
                return p.device
            for b in self.buffers():
                return b.device
            return torch.device("cpu")


def require_gpu(t: torch.Tensor, who: str):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: input is on {t.device}; tricolo_amd runs only on an MI355X (HIP extension, no CPU "
                           f"fallback).  Use the oracle/ package for CPU reference arithmetic.")


_linear_geoms: dict = {}


def linear_geom(rows: int, K: int, N: int, spatial: int = 1) -> ops.ConvGeom:
    """Linear(K*spatial -> N) as a conv over a `spatial`-site grid: x is channels-last [rows, spatial, K] and the
    parameter is the reference's [N, K*spatial] with input index = c*spatial + s (channels-first flatten of
    sparse_cnn.py:49), i.e. strides (s_co, s_tap, s_ci) = (K*spatial, 1, spatial)."""
    key = (rows, K, N, spatial)
    g = _linear_geoms.get(key)
    if g is None:
        if spatial == 1:
            g = ops.ConvGeom(rows, (1, 1, 1), K, K, N, (1, 1, 1), 1, (0, 0, 0), (K, 1, 1))
        else:
            e = round(spatial ** (1 / 3))
            assert e ** 3 == spatial
            g = ops.ConvGeom(rows, (e, e, e), K, K, N, (e, e, e), e, (0, 0, 0), (K * spatial, 1, spatial))
        _linear_geoms[key] = g
    return g


_NO_SMALL = False                                               # (round 6: the A/B switch of the <= 64-row dense kernels was dropped)


def _small(x, w, spatial) -> bool:
    return (not _NO_SMALL and spatial == 1 and x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous() and w.dim() == 2
            and ops.linear_small_supported(x.shape[0], w.shape[1], w.shape[0]))


def linear_fwd(x, w, b, act: int, precision: str, spatial: int = 1):
    """x [rows, spatial*K] channels-last -> act(x W^T + b) [rows, N] on the MFMA conv kernel (act 0/1 relu/2 tanh)."""
    precision = ops.head_precision(precision)
    rows = x.shape[0]
    N = w.shape[0]
    K = w.shape[1] // spatial
    if _small(x, w, spatial):
        return ops.linear_small_fwd(x, w, b, act, precision)        # <= 64 rows: one launch, no packed copy of W
    g = linear_geom(rows, K, N, spatial)
    packed = ops.pack_weight(w, g, precision)
    out = ops.conv_fwd(x, g, packed, bias=b, act=act)
    return out.view(rows, N)


def linear_bwd(x, w, out, dout, act: int, precision: str, spatial: int = 1, need_dx: bool = True, need_db: bool = True):
    """Backward of linear_fwd.  Returns (dx | None, dw, db)."""
    precision = ops.head_precision(precision)
    rows = x.shape[0]
    N = w.shape[0]
    K = w.shape[1] // spatial
    if _small(x, w, spatial) and w.dim() == 2:
        return ops.linear_small_bwd(x, w, out, dout, act, precision, need_dx, need_db)
    g = linear_geom(rows, K, N, spatial)
    dpre = ops.act_bwd(dout.contiguous().clone(), out, act) if act else dout.contiguous()
    dw = ops.conv_wgrad(x, dpre, g, w, precision)
    db = ops.colsum(dpre) if need_db else None
    dx = None
    if need_dx:
        packed_t = ops.pack_weight(w, g, precision, transposed=True)
        dx = ops.conv_dgrad(dpre, g, packed_t).view(x.shape)
    return dx, dw, db


class LinearFn(torch.autograd.Function):
    """Autograd wrapper of one dense layer (used where a tower is not one fused Function: CLIP-text MLP, BiGRU fc)."""

    @staticmethod
    def forward(ctx, x, w, b, act, precision):
        x = x.contiguous()
        out = linear_fwd(x, w, b, act, precision)
        ctx.save_for_backward(x, w, out)
        ctx.act, ctx.precision = act, precision
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, out = ctx.saved_tensors
        dx, dw, db = linear_bwd(x, w, out, dout, ctx.act, ctx.precision, need_dx=ctx.needs_input_grad[0])
        return dx, dw, db, None, None


class L2NormFn(torch.autograd.Function):
    """F.normalize(x, dim=1) as one kernel each way."""

    @staticmethod
    def forward(ctx, x):
        z, norm = ops.l2norm_fwd(x.contiguous())
        ctx.save_for_backward(z, norm)
        return z

    @staticmethod
    def backward(ctx, dz):
        z, norm = ctx.saved_tensors
        return ops.l2norm_bwd(z, norm, dz)


class SideStream:
    """Weight-gradient side stream of a tower.  wgrad(layer) and dgrad(layer) both consume the same dy and are
    independent, and each leaves most CUs idle (latency-bound k-loops, few workgroups), so the tower's wgrad launches
    run on a second HIP stream next to the dgrad / BatchNorm-backward chain and are joined at the end of the backward.
    The fork/join is captured into the HIP graph as parallel branches."""

    def __init__(self, tag=""):
        self.stream = None
        self.enabled = tag not in os.environ.get("TRICOLO_NO_SIDE", "").split(",")

    def fork(self, *tensors, event=None):
        """Make the side stream wait for everything issued so far on the current stream (or, with `event`, for what had been issued when
        that event was recorded: the branch is then ISSUED here but depends on less); returns the side stream."""
        if not self.enabled:
            return torch.cuda.current_stream()
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        if event is not None:
            self.stream.wait_event(event)
        else:
            self.stream.wait_stream(torch.cuda.current_stream())
        if not torch.cuda.is_current_stream_capturing():   # (under capture the graph's private pool defers every free)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.stream)      # allocator: do not recycle these while the side stream reads them
        return self.stream

    def join(self, *tensors):
        if self.stream is not None:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self.stream)
            if not torch.cuda.is_current_stream_capturing():
                for t in tensors:
                    if t is not None:
                        t.record_stream(cur)
