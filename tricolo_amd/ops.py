"""Tensor-level wrappers over the C ABI (one Python function per kernel family).

Precision modes: "bf16x3" fp32 activation storage, 3-product split bf16 operands (strict parity);
"f16" f16 activation storage and operands, fp32 accumulation, bf16x3 heads / GRU (inside the 1e-3 parity bound at the
speed of the bf16 mode); "bf16" bf16 storage and operands (throughput mode of BASELINE config 2).

All tensors are contiguous, channels-last ([B, D, H, W, C]; 2D tensors carry D = 1) and live on the GPU.
Nothing here falls back to torch arithmetic: every function launches a kernel of libtricolo_hip.so on the
current torch stream (so the calls can be captured into a HIP graph together with the rest of the step).
"""
from __future__ import annotations

import math
import os

import torch

from . import _C
from ._C import check, lib, make_desc, ptr, stream

_PRECISIONS = ("bf16", "bf16x3", "f16")
_FMT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}        # act_fmt of include/tricolo_hip.h
# Gradients of the big activation tensors are stored in f16 in the f16 mode.  Their magnitude shrinks with the number of
# positions they spread over (1e-6 .. 1e-5 per element in layer1 at batch 32 x 6 views), below f16's normal range
# (6.1e-5): they are carried multiplied by 2^12 from the first 16-bit gradient tensor of a tower to the parameter-gradient
# kernels, which multiply by 2^-12 (exact).  f16 max is 65504.
F16_GRAD_SCALE = float(os.environ.get("TRICOLO_F16_GRAD_SCALE", "4096"))      # (a power of two: the un-scaling is exact)
if not (F16_GRAD_SCALE >= 1.0 and math.frexp(F16_GRAD_SCALE)[0] == 0.5):
    raise ValueError(f"TRICOLO_F16_GRAD_SCALE={F16_GRAD_SCALE!r}: must be a power of two >= 1 (the kernels divide it out exactly)")
_default_precision = os.environ.get("TRICOLO_PRECISION", "bf16x3")


def set_default_precision(p: str):
    global _default_precision
    if p not in _PRECISIONS:
        raise ValueError(f"precision must be one of {_PRECISIONS}")
    _default_precision = p


def default_precision() -> str:
    return _default_precision


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg).  Disabled by
    default; when enabled every conv launch is bracketed by two events and tallied per kernel symbol."""

    def __init__(self):
        self.records = []          # (symbol, algorithmic_flops, start_event, end_event)
        self.overhead_ms = 0.0     # fixed cost of one event pair around a launch (calibrate()), subtracted per record

    def calibrate(self, n=96):
        """Event-pair time of an (almost) empty kernel launched the same way: what a bracketed launch pays on top of its kernel's own
        duration (event processing + the dispatch gap; ~1.3 us of it is the stamp kernel itself and is left in).  Call behind a
        spin kernel so the launches are queued back to back like the step's."""
        buf = torch.zeros(1, dtype=torch.int64, device="cuda")
        pairs = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            check(lib().tri_debug_stamp(buf.data_ptr(), stream()), "tri_debug_stamp")
            b.record()
            pairs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in pairs)
        self.overhead_ms = max(ts[len(ts) // 2] - 1.3e-3, 0.0)
        return self.overhead_ms

    def run(self, symbol, flops, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.records.append((symbol, flops, a, b))
        return r

    def summary(self):
        """Per symbol: launches, summed event time (ms: the calibrated event-pair overhead subtracted per launch; ms_raw: as measured),
        flops = EXECUTED FLOPs (launches over a compact row list or a site mask count their active rows - resolved here from the
        device-side counts, after the synchronize), flops_dense = the dense-equivalent figure beside it."""
        torch.cuda.synchronize()
        agg = {}
        for sym, fl, a, b in self.records:
            d = agg.setdefault(sym, {"launches": 0, "ms": 0.0, "flops": 0, "flops_dense": 0})
            d["launches"] += 1
            e = a.elapsed_time(b)
            d["ms_raw"] = d.get("ms_raw", 0.0) + e
            d["ms"] += max(e - self.overhead_ms, 0.25 * e)
            ex, dense = fl() if callable(fl) else (fl, fl)
            d["flops"] += ex
            d["flops_dense"] += dense
        return agg


TIMER: KernelTimer | None = None


def _timed(symbol, flops, fn):
    return TIMER.run(symbol, flops, fn) if TIMER is not None else fn()


def _flops(g, rows=None, row_mask=None):
    """FLOPs of one launch of layer g for the KernelTimer: dense 2*M*taps*Cin*Cout, or - for launches that contract / compute over a
    compact row list or a site mask - a callable -> (executed, dense) with executed = 2 * active rows * taps * Cin * Cout, resolved at
    summary time from the device-side count (masked launches are priced on executed FLOPs)."""
    if TIMER is None or (rows is None and row_mask is None):
        return g.flops
    per_row = 2 * g.ntaps * g.cin * g.cout
    n = g.M
    if rows is not None:
        cnt = rows[1]
        return lambda: (per_row * min(int(cnt.item()), n), g.flops)
    return lambda: (per_row * int(row_mask[:n].sum().item()), g.flops)


def _flops_sum(items):
    """Sum of _flops() values (numbers or callables) as one value of the same kind."""
    if not any(callable(f) for f in items):
        return sum(items)

    def total():
        ex = dn = 0
        for f in items:
            e, d = f() if callable(f) else (f, f)
            ex, dn = ex + e, dn + d
        return ex, dn
    return total


_TNAME = {torch.float32: "float", torch.bfloat16: "bf16", torch.float16: "f16"}


def _igemm_symbol(g, transposed, split, t):
    """Name of the kernel the C dispatch picks for this layer (what rocprofv3 reports), for the KernelTimer."""
    h16 = t.dtype != torch.float32
    code = g.kernel_family[(transposed, 1 if split else (2 if h16 else 0))]
    fam, bn = code & 255, (code >> 8) & 255
    if fam == 6:
        return f"conv_vox0_kernel<{_TNAME[t.dtype]}>"
    if fam == 14:
        return f"conv_voxb_kernel<{_TNAME[t.dtype]}>"
    if fam == 13:
        return f"conv_voxg_kernel<{bn}, {_TNAME[t.dtype]}>"                # (forward and data gradient under one name)
    if fam == 9:
        return f"conv_c64_kernel<{_TNAME[t.dtype]}>"                 # (forward / data gradient (+ accumulate) under one name)
    if fam == 10:
        return f"conv_s2d_kernel<{_TNAME[t.dtype]}>"
    if fam == 15:
        return f"conv_s2g_kernel<{_TNAME[t.dtype]}>"
    if fam == 12:
        return f"conv_pw_kernel<{bn}, {_TNAME[t.dtype]}>"                # (forward and data gradient under one name)
    if fam == 4:
        return f"conv_stem_kernel<{g.kernel[1]}, {_TNAME[t.dtype]}>"
    if fam == 5:
        return f"conv_halo_rows_kernel<{_TNAME[t.dtype]}>"          # (store / accumulate variants under one name)
    if fam == 3:
        return f"conv_halo2d_kernel<{bn}, {_TNAME[t.dtype]}>"     # (both record variants under one name)
    if fam == 2:
        return f"conv_dma_kernel<{bn}, {_TNAME[t.dtype]}>"           # (pipeline depths 2 / 3 / 4 under one name)
    return f"conv_igemm_kernel<{bn}, {2 if split else 1}, {_TNAME[t.dtype]}>"


def _f32(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), "expected contiguous fp32"
    return t


def _act(t):
    """Activation tensor: contiguous fp32 (bf16x3 mode), bf16 (bf16 mode) or f16 (f16 mode)."""
    assert t.dtype in _FMT and t.is_contiguous(), "expected contiguous fp32 / bf16 / f16 activations"
    return t


def _abf(t) -> int:
    """act_fmt of a tensor: 0 fp32, 1 bf16, 2 f16."""
    return _FMT[t.dtype]


def act_dtype(precision: str):
    """Storage type of the big activation tensors for a precision mode (TRICOLO_ACT_FP32=1 forces fp32 storage in bf16 mode)."""
    if precision == "f16":
        return torch.float16
    if precision == "bf16" and os.environ.get("TRICOLO_ACT_FP32", "0") != "1":
        return torch.bfloat16
    return torch.float32


def split3(precision: str) -> int:
    """3-product split bf16 operands in the fp32 kernels (heads, GRU): the strict mode and the f16 mode."""
    return 1 if precision in ("bf16x3", "f16") else 0


def head_precision(precision: str) -> str:
    """Precision of the dense fp32 layers (MLP heads, GRU input projection): the f16 mode runs them in bf16x3."""
    return "bf16x3" if precision == "f16" else precision


def grad_scale(precision: str) -> float:
    return F16_GRAD_SCALE if precision == "f16" else 1.0


def _conv_mode(x, lo) -> int:
    """Plan index of the conv entry points: 0 bf16 operands / fp32 storage, 1 bf16x3, 2 16-bit storage."""
    return 1 if lo is not None else (2 if x.dtype != torch.float32 else 0)


_FRAG_FAMILIES = (13, 14, 15)        # kernel families that read the packed operand fragment-major: conv_voxg / conv_voxb / conv_s2g


class ConvGeom:
    """Geometry + packed-weight buffers of one conv / linear layer.

    weight_strides = (s_co, s_tap, s_ci) element strides of the fp32 parameter in the reference's layout."""

    def __init__(self, B, in_grid, cin, cin_stored, cout, kernel, stride, pad, weight_strides):
        ID, IH, IW = in_grid
        KD, KH, KW = kernel
        pd, ph, pw = pad
        OD = (ID + 2 * pd - KD) // stride + 1
        OH = (IH + 2 * ph - KH) // stride + 1
        OW = (IW + 2 * pw - KW) // stride + 1
        self.B, self.in_grid, self.out_grid = B, (ID, IH, IW), (OD, OH, OW)
        self.cin, self.cin_stored, self.cout = cin, cin_stored, cout
        self.kernel, self.stride, self.pad = kernel, stride, pad
        self.ntaps = KD * KH * KW
        self.strides = weight_strides
        self.desc = make_desc(B, ID, IH, IW, cin_stored, OD, OH, OW, cout, KD, KH, KW, stride, pd, ph, pw)
        self.kpad = lib().tri_conv_kpad(self.ntaps, cin_stored)
        self.kpad_t = lib().tri_conv_kpad(self.ntaps, cout)
        self.M = B * OD * OH * OW
        self.M_in = B * ID * IH * IW
        self.num_mtiles = {m: lib().tri_conv_num_mtiles(_C.C.byref(self.desc), m) for m in (0, 1, 2)}     # by _conv_mode
        self.num_records_rows = {m: lib().tri_conv_num_records(_C.C.byref(self.desc), m, 1) for m in (0, 1, 2)}   # calls with a row list
        self.kernel_family = {(tr, m): lib().tri_conv_kernel_family(_C.C.byref(self.desc), 1 if tr else 0, m)
                              for tr in (False, True) for m in (0, 1, 2)}
        self.wgrad_ws = lib().tri_conv_wgrad_workspace(_C.C.byref(self.desc))
        fam = lib().tri_conv_wgrad_kernel_family(_C.C.byref(self.desc), 1)
        self.wgrad_dma = fam == 2
        self.wgrad_krow = fam == 7                                # conv_wgrad_krow_kernel (resolution-keeping 3x3 layers, 16-bit storage, no row mask / list)
        self.wgrad_brick = fam == 6                               # conv_vox0_wgrad_kernel: needs the dense site mask (row_mask), not a row list
        self._plans = {}
        self.fwd_ws = lib().tri_conv_workspace(_C.C.byref(self.desc), 0)
        self.dgrad_ws = lib().tri_conv_workspace(_C.C.byref(self.desc), 1) if cin == cin_stored and cin % 32 == 0 else 0

    def dgrad_row_order(self, device):
        """Stride-2 layers: the input positions sorted by the parity class of (coordinate + pad) per axis - the order in which
        tri_conv_dgrad should visit its rows so that a tile only runs the taps its rows can use.  None for stride 1."""
        if self.stride != 2 or self.ntaps == 1:                  # 1x1 / 2: one tap, nothing to skip (and scattered rows cost)
            return None
        t = self._plans.get(("rowpos", device))
        if t is None:
            ID, IH, IW = self.in_grid
            pd, ph, pw = self.pad
            pos = torch.arange(self.M_in, dtype=torch.int64)
            x, r = pos % IW, pos // IW
            y, r = r % IH, r // IH
            z = r % ID
            cls = (((z + pd) & 1) << 2) | (((y + ph) & 1) << 1) | ((x + pw) & 1)
            order = torch.sort(cls * self.M_in + pos).indices          # by class, ascending position inside a class
            t = order.to(torch.int32).to(device)
            self._plans[("rowpos", device)] = t
        return t

    def brick(self, transposed: bool, mode: int) -> bool:
        """True when tri_conv_fwd / tri_conv_dgrad runs this layer on a kernel that walks the dense grid by the SITE MASK (the brick kernels
        of conv_vox.hip, conv_voxg_kernel on the coarse grids): such launches take the mask as row_mask (rows of inactive sites are then
        neither computed nor written), not a compact row list."""
        return (self.kernel_family[(transposed, mode)] & 255) in (6, 7, 13, 14)

    def packed_frag(self, transposed: bool, precision: str, storage=None) -> int:
        """1 when the kernel that consumes this layer's packed operand (forward / data gradient) reads it in MFMA-fragment-major order
        (tri_weight_prep frag: conv_voxg_kernel loads its weight fragments straight into registers).  The plan mode follows the
        activation storage the operand will meet (default: the precision mode's own)."""
        storage = storage or act_dtype(precision)
        mode = 1 if precision == "bf16x3" else (2 if storage != torch.float32 else 0)
        return 1 if (self.kernel_family[(transposed, mode)] & 255) in _FRAG_FAMILIES else 0

    def check_packed(self, packed, transposed: bool, x):
        """The packed operand must be in the order this call's kernel reads (see pack_weight's `storage`)."""
        need = 1 if (self.kernel_family[(transposed, _conv_mode(x, packed[1]))] & 255) in _FRAG_FAMILIES else 0
        if getattr(packed[0], "tri_frag", 0) != need:
            raise RuntimeError("conv: the packed operand was ordered for another plan (row-major vs fragment-major): pack it with "
                               "pack_weight(..., storage=<dtype of the activations>)")

    def voxg_spu(self, transposed: bool, mode: int) -> int:
        """Samples per unit of conv_voxg_kernel for this layer (0: another kernel runs it)."""
        code = self.kernel_family[(transposed, mode)]
        return (code >> 24) & 127 if (code & 255) == 13 else 0

    def splitk(self, transposed: bool, mode: int) -> bool:
        """True when tri_conv_fwd / tri_conv_dgrad runs this layer split-K in plan `mode` (_conv_mode): such launches take
        row_mask, not a compact row list."""
        return bool(self.kernel_family[(transposed, mode)] >> 16 & 1)

    def dgrad_bn_records(self, accumulate: bool, fmt: int) -> int:
        """Records tri_conv_dgrad_bn writes for this layer (0: its data-gradient kernel has no fused BatchNorm-backward sums), cached."""
        c = self.__dict__.setdefault("_dgrad_bn", {})
        k = (bool(accumulate), fmt)
        if k not in c:
            c[k] = int(lib().tri_conv_dgrad_bn_records(_C.C.byref(self.desc), 1 if accumulate else 0, fmt))
        return c[k]

    def wgrad_group(self, fmt: int):
        """(family, tiles, steps) of this layer's weight gradient for WgradBatch.add_job (tri_conv_wgrad_group_info), cached."""
        c = self.__dict__.setdefault("_wgrad_group", {})
        if fmt not in c:
            fam, tiles, steps = _C.C.c_int(), _C.C.c_int(), _C.C.c_int()
            check(lib().tri_conv_wgrad_group_info(_C.C.byref(self.desc), fmt, _C.C.byref(fam), _C.C.byref(tiles), _C.C.byref(steps)),
                  "tri_conv_wgrad_group_info")
            c[fmt] = (fam.value, tiles.value, steps.value)
        return c[fmt]

    def plan(self, device):
        """Gather plan (built once per geometry and device, reused by every step's wgrad)."""
        pl = self._plans.get(device)
        if pl is None:
            pl = torch.empty((lib().tri_conv_plan_bytes(_C.C.byref(self.desc)),), dtype=torch.uint8, device=device)
            check(lib().tri_conv_plan_build(_C.C.byref(self.desc), ptr(pl), stream()), "tri_conv_plan_build")
            self._plans[device] = pl
        return pl

    @property
    def flops(self):
        return 2 * self.M * self.ntaps * self.cin * self.cout


def pack_weight(w: torch.Tensor, g: ConvGeom, precision: str, transposed: bool = False, storage=None):
    """fp32 parameter -> (hi, lo|None) bf16 MFMA operand rows.  transposed=True packs the dgrad operand.  storage: dtype of the activation
    tensors the operand will meet (default act_dtype(precision)) - it selects the plan mode and with it the operand ORDER (row-major, or
    fragment-major for conv_voxg_kernel); conv_fwd / conv_dgrad refuse an operand packed for another plan."""
    s_co, s_tap, s_ci = g.strides
    if not transposed:
        rows, inner, inner_pad, kpad, s_row, s_inner = g.cout, g.cin, g.cin_stored, g.kpad, s_co, s_ci
    else:
        rows, inner, inner_pad, kpad, s_row, s_inner = g.cin_stored, g.cout, g.cout, g.kpad_t, s_ci, s_co
        if g.cin != g.cin_stored:
            raise RuntimeError("dgrad operand requested for a layer with padded input channels")
    hi = torch.empty((rows, kpad), dtype=torch.float16 if precision == "f16" else torch.bfloat16, device=w.device)
    lo = torch.empty_like(hi) if precision == "bf16x3" else None
    frag = g.packed_frag(transposed, precision, storage)
    check(lib().tri_weight_prep(ptr(_f32(w)), s_row, s_tap, s_inner, rows, g.ntaps, inner, inner_pad, ptr(hi), ptr(lo), _abf(hi),
                                frag, stream()), "tri_weight_prep")
    hi.tri_frag = frag
    return hi, lo


class WeightPacker:
    """Packs the MFMA operand rows of every layer of a tower in ONE launch per step (forward + data-gradient operands).

    Entries are registered once (parameter, geometry, direction); the packed buffers and the device-side descriptor
    table are persistent, so a step costs one kernel instead of one per layer.  The table is rebuilt if a parameter's
    storage moved (e.g. after FusedAdam re-bound the parameters to its flat buffer)."""

    def __init__(self):
        self.entries = {}            # key -> (param, geom, transposed)
        self._tables = {}            # precision -> (desc tensor, n, {key: (hi, lo)}, ptr snapshot)

    def add(self, key, param, g: "ConvGeom", transposed: bool = False):
        if key not in self.entries:
            self.entries[key] = (param, g, transposed)
            self._tables.clear()

    def _build(self, precision, device):
        descs = (_C.TriPrepDesc * len(self.entries))()
        bufs = {}
        for i, (key, (w, g, tr)) in enumerate(self.entries.items()):
            s_co, s_tap, s_ci = g.strides
            if not tr:
                rows, inner, inner_pad, kpad, s_row, s_inner = g.cout, g.cin, g.cin_stored, g.kpad, s_co, s_ci
            else:
                rows, inner, inner_pad, kpad, s_row, s_inner = g.cin_stored, g.cout, g.cout, g.kpad_t, s_ci, s_co
            hi = torch.empty((rows, kpad), dtype=torch.float16 if precision == "f16" else torch.bfloat16, device=device)
            lo = torch.empty_like(hi) if precision == "bf16x3" else None
            bufs[key] = (hi, lo)
            d = descs[i]
            d.fmt = _abf(hi)
            d.w, d.hi, d.lo = w.data_ptr(), hi.data_ptr(), (lo.data_ptr() if lo is not None else None)
            d.s_row, d.s_tap, d.s_inner = s_row, s_tap, s_inner
            d.rows, d.ntaps, d.inner, d.inner_pad, d.kpad = rows, g.ntaps, inner, inner_pad, kpad
            d.frag = hi.tri_frag = g.packed_frag(tr, precision)
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device)
        snap = tuple(w.data_ptr() for (w, _, _) in self.entries.values())
        return raw, len(self.entries), bufs, snap

    def run(self, precision, device):
        """Launch the packing of all registered layers; returns {key: (hi, lo)}."""
        tab = self._tables.get(precision)
        if tab is None or tab[3] != tuple(w.data_ptr() for (w, _, _) in self.entries.values()):
            tab = self._build(precision, device)
            self._tables[precision] = tab
        raw, n, bufs, _ = tab
        check(lib().tri_weight_prep_multi(ptr(raw), n, stream()), "tri_weight_prep_multi")
        return bufs


def conv_fwd(x, g: ConvGeom, packed, row_mask=None, bias=None, act=0, want_stats=False, out=None, accumulate=False, rows=None):
    """rows = (row_pos, count) from mask_compact: only those rows are computed and WRITTEN (submanifold layers; pass no row_mask)."""
    hi, lo = packed
    g.check_packed(packed, False, x)
    OD, OH, OW = g.out_grid
    if out is None:
        out = torch.empty((g.B, OD, OH, OW, g.cout), dtype=x.dtype, device=x.device)
    assert out.dtype == x.dtype
    nrec = (g.num_records_rows if rows else g.num_mtiles)[_conv_mode(x, lo)]
    stats = torch.empty((nrec, 2, g.cout), dtype=torch.float32, device=x.device) if want_stats else None
    ws = _workspace(g.fwd_ws, x.device) if g.fwd_ws else None
    check(_timed(_igemm_symbol(g, False, lo is not None, x), _flops(g, rows, row_mask),
                 lambda: lib().tri_conv_fwd(_C.C.byref(g.desc), ptr(_act(x)), ptr(hi), ptr(lo), ptr(out), ptr(row_mask), ptr(bias),
                                            act, 1 if accumulate else 0, ptr(stats), _abf(x), ptr(ws),
                                            ws.numel() if ws is not None else 0, ptr(rows[0]) if rows else None,
                                            ptr(rows[1]) if rows else None, stream())), "tri_conv_fwd")
    return (out, stats) if want_stats else out


_ROW_ORDER = True                                                    # parity-class row order for stride-2 data gradients (A/B switch dropped in round 6)


# A/B switch: 0 = BatchNorm-backward sums as a pass of their own everywhere, 2 (default) = only the relu(bn1) form inside conv2's data
# gradient, 1 = also the relu(bn2 + x) form inside the accumulated conv1 gradient (reads y AND the saved output cold: 40.0 us against
# 38.3 us for data gradient + reduce pass, tools/dgrad_bn_bench.py; the relu(bn1) form 27.0 against 31.2)
_DGRAD_BN_MODE = os.environ.get("TRICOLO_DGRAD_BN", "2")
_DGRAD_BN = _DGRAD_BN_MODE != "0"


def conv_dgrad(dout, g: ConvGeom, packed_t, row_mask=None, out=None, accumulate=False, rows=None, bn_sums=None):
    """Data gradient.  bn_sums = (y, BNCoeffs | None, relu_out | None): the caller's next pass over the result is the BatchNorm-backward
    reduce of the BatchNorm that consumed y; where the layer's kernel can take those sums in its epilogue (tri_conv_dgrad_bn_records)
    the call returns (din, partial) and bn_bwd(..., partial=partial) skips its reduce launch, otherwise (din, None)."""
    hi, lo = packed_t
    g.check_packed(packed_t, True, dout)
    ID, IH, IW = g.in_grid
    if out is None:
        out = torch.empty((g.B, ID, IH, IW, g.cin_stored), dtype=dout.dtype, device=dout.device)
    assert out.dtype == dout.dtype
    ws = _workspace(g.dgrad_ws, dout.device) if g.dgrad_ws else None
    rowpos = g.dgrad_row_order(dout.device) if (row_mask is None and rows is None and _ROW_ORDER) else None
    if rows is not None:
        rowpos = rows[0]
    if bn_sums is not None:
        y, co, relu_out = bn_sums
        nrec = g.dgrad_bn_records(accumulate, _abf(dout)) if (_DGRAD_BN and not (accumulate and _DGRAD_BN_MODE == "2") and row_mask is None and rows is None and lo is None
                                                               and (relu_out is not None) == bool(accumulate)
                                                               and (co is not None) != (relu_out is not None)) else 0
        if nrec == 0:
            return conv_dgrad(dout, g, packed_t, row_mask=row_mask, out=out, accumulate=accumulate, rows=rows), None
        assert y.dtype == out.dtype and y.numel() == out.numel()
        partial = torch.empty((nrec, 2, g.cin_stored), dtype=torch.float32, device=dout.device)
        sums = _C.TriConvBnSums(ptr(y), ptr(co.scale) if co is not None else None, ptr(co.shift) if co is not None else None,
                                ptr(relu_out), ptr(partial))
        check(_timed(_igemm_symbol(g, True, False, dout), g.flops,
                     lambda: lib().tri_conv_dgrad_bn(_C.C.byref(g.desc), ptr(_act(dout)), ptr(hi), None, ptr(out), 1 if accumulate else 0,
                                                     _abf(dout), ptr(ws), ws.numel() if ws is not None else 0, ptr(rowpos),
                                                     _C.C.byref(sums), stream())), "tri_conv_dgrad_bn")
        return out, partial
    check(_timed(_igemm_symbol(g, True, lo is not None, dout), _flops(g, rows, row_mask),
                 lambda: lib().tri_conv_dgrad(_C.C.byref(g.desc), ptr(_act(dout)), ptr(hi), ptr(lo), ptr(out), ptr(row_mask),
                                              1 if accumulate else 0, _abf(dout), ptr(ws), ws.numel() if ws is not None else 0,
                                              ptr(rowpos), ptr(rows[1]) if rows else None, stream())), "tri_conv_dgrad")
    return out


_wgrad_ws = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _wgrad_ws[key] = ws
    return ws


_WGRAD_GROUPED = os.environ.get("TRICOLO_WGRAD_GROUPED", "1") != "0"     # A/B switch: one reduce launch per layer instead
_WGRAD_JOBS = os.environ.get("TRICOLO_WGRAD_JOBS", "1") != "0"           # A/B switch: one partial launch per layer instead
# output tiles (workgroups per split) one grouped partial launch takes, per kernel family (1: 128-row, 2: 64-row, 3: 256-row tiles); the
# library's split planner uses the same figures
# (4, 5: conv_wgrad_krow_kernel<128> / <64>; a krow launch may hold more workgroups than resident slots, the library deals them longest first)
_WGRAD_JOB_TILES = dict(zip(range(1, 8), [512, 448, 256, 512, 512, 512, 512]))   # (6, 7: stride-2 krow; the tuning aid for these figures was dropped in round 6)


_WGRAD_REDUCE_OVERLAP = os.environ.get("TRICOLO_WGRAD_REDUCE_OVERLAP", "1") != "0"   # A/B switch (round 6): WgradBatch.flush(side=...)


# Overflow note (round 6): the optimizer whose zero_grad() ran last lends the grouped weight-gradient reduce its device record; the reduce flags
# a step that STORED an inf / NaN exactly as the optimizer's own scan would, and tags every gradient tensor it vouched for (`_tri_noted` =
# (address of the record, the tensor's version counter)) - FusedAdam leaves those out of its scan (64 of the default step's 74 MB of
# gradients).  The tag travels with the tensor OBJECT: a gradient that reaches the optimizer as another tensor (accumulated over several
# backward passes, cloned by AccumulateGrad, made contiguous) has no tag, one that was modified in place no longer matches its version -
# both are scanned as before.  (No table of addresses, and no reference to the tensors is kept here: a second owner makes autograd's
# AccumulateGrad CLONE every gradient it is handed - measured: +120 us per step.)
_GUARD_NOTE = None           # int32[4] device tensor (kept alive here) or None


def set_guard_note(note):
    global _GUARD_NOTE
    _GUARD_NOTE = note


def guard_noted(g, note) -> bool:
    """True when g's writer already noted its inf / NaN in `note` and nobody touched g since.  Consumes the tag: a second optimizer step on
    the same tensor scans it."""
    tag = g.__dict__.pop("_tri_noted", None)
    return tag is not None and tag == (note.data_ptr(), g._version)


class WgradBatch:
    """Deferred weight-gradient reduces of one tower backward (tri_conv_wgrad_partial / tri_wgrad_reduce_grouped).

    Every conv_wgrad(..., batch=b) launches only the position-split partial kernel, into a slab of its own carved from a
    per-stream arena that persists across steps (so a captured HIP graph replays on the same memory); b.flush() sums all of
    them in one launch.  The returned dw tensors hold garbage until then: flush before they leave the autograd Function.

    Layers of one kernel family (tri_conv_wgrad_group_info) are not even launched one by one: their jobs wait here until a launch's
    worth of output tiles (~448 workgroups) or TRI_WGRAD_JOBS_MAX of them are pending and then share ONE partial launch
    (tri_conv_wgrad_partial_group) - each layer is cut into fewer, longer splits, so the fp32 slabs the reduce has to re-read
    shrink by about the number of jobs.  The job keeps x / dout alive until that launch."""
    _arenas = {}

    def __init__(self, device, group_jobs: bool | None = None):
        self.device = device
        self.group_jobs = _WGRAD_JOBS if group_jobs is None else bool(group_jobs)
        # arenas of a HIP-graph capture are kept apart from the eager ones of the same stream id: a chunk allocated under capture lives
        # in the graph's private pool and must only ever be touched by replays (torch hands stream ids out of a small pool, so an
        # eager step could otherwise land on a capture's arena - ADVICE r2)
        self.chunks = WgradBatch._arenas.setdefault((device, torch.cuda.current_stream().cuda_stream, torch.cuda.is_current_stream_capturing()), [])
        self.ci, self.off, self.descs = 0, 0, []
        self.queues = {}                                          # kernel family -> [pending jobs, their output tiles]
        self._pre = None                                          # prelaunch() state: (descs to reduce early, event, family left for flush)
        self.dws = {}                                             # data_ptr -> gradient tensor of every layer queued here (overflow note)

    @property
    def jobs(self):
        return [j for q in self.queues.values() for j in q[0]]

    def add_job(self, family, tiles, job, keep, flops, sym):
        """Queue one layer's partial kernel in its family's queue (see the class docstring); that queue is launched first when this
        job does not fit beside what is pending."""
        q = self.queues.setdefault(family, [[], 0])
        if q[0] and (q[1] + tiles > _WGRAD_JOB_TILES[family] or len(q[0]) == _C.TRI_WGRAD_JOBS_MAX):
            self.launch_jobs(family)
            q = self.queues.setdefault(family, [[], 0])
        q[0].append((job, keep, flops, sym))
        q[1] += tiles

    def launch_jobs(self, family=None):
        for fam in ([family] if family is not None else sorted(self.queues)):
            pending = self.queues.pop(fam, [[], 0])[0]
            n = len(pending)
            if not n:
                continue
            arr = (_C.TriWgradJob * n)(*[j[0] for j in pending])
            pend = (_C.TriWgradReduce * n)()
            fmt = _abf(pending[0][1][0])
            check(_timed(pending[0][3], _flops_sum([j[2] for j in pending]),
                         lambda: lib().tri_conv_wgrad_partial_group(arr, n, fmt, pend, stream())), "tri_conv_wgrad_partial_group")
            for i in range(n):
                d = _C.TriWgradReduce()
                _C.C.memmove(_C.C.byref(d), _C.C.byref(pend[i]), _C.C.sizeof(d))
                self.descs.append(d)

    def slab(self, nbytes: int) -> torch.Tensor:
        nbytes = (nbytes + 255) // 256 * 256
        while True:
            if self.ci < len(self.chunks):
                if self.off + nbytes <= self.chunks[self.ci].numel():
                    t = self.chunks[self.ci][self.off:self.off + nbytes]
                    self.off += nbytes
                    return t
                self.ci, self.off = self.ci + 1, 0
            else:
                self.chunks.append(torch.empty(max(nbytes, 16 << 20), dtype=torch.uint8, device=self.device))

    @staticmethod
    def reserve(device, nbytes: int):
        """Make sure the current stream's arena holds at least nbytes (call eagerly, ahead of a capture)."""
        chunks = WgradBatch._arenas.setdefault((device, torch.cuda.current_stream().cuda_stream, torch.cuda.is_current_stream_capturing()), [])
        have = sum(c.numel() for c in chunks)
        if have < nbytes:
            chunks.append(torch.empty(max(nbytes - have, 16 << 20), dtype=torch.uint8, device=device))

    @staticmethod
    def release():
        """Drop every arena (between runs / modes: the chunks are plain torch allocations)."""
        WgradBatch._arenas.clear()

    def _reduce(self, descs):
        n = len(descs)
        if n:
            arr = (_C.TriWgradReduce * n)(*descs)
            note = _GUARD_NOTE
            if note is not None and self.device.index is not None and note.device.index != self.device.index:
                note = None
            check(lib().tri_wgrad_reduce_grouped_noted(arr, n, ptr(note), stream()), "tri_wgrad_reduce_grouped")
            if note is not None:
                for d in descs:
                    t = self.dws.pop(int(d.dw or 0), None)
                    if t is not None:
                        t._tri_noted = (note.data_ptr(), t._version)

    def prelaunch(self):
        """First half of flush(side=...): every pending family but the last is launched now (smallest first, the two largest swapped - see
        flush).  The caller may issue other work between this and flush(); -> True when something is left for flush() to overlap."""
        if self._pre is not None:
            return True
        fams = [f for f in sorted(self.queues) if self.queues[f][0]]
        if not _WGRAD_REDUCE_OVERLAP or len(fams) < 2:
            return False

        def work(f):                                              # dense layers carry their FLOPs as numbers (row-list layers: callables)
            fl = [j[2] for j in self.queues[f][0]]
            return sum(v for v in fl if not callable(v)) + 1e9 * self.queues[f][1] * any(callable(v) for v in fl)
        fams.sort(key=work)
        fams[-1], fams[-2] = fams[-2], fams[-1]
        for f in fams[:-1]:
            self.launch_jobs(f)
        early, self.descs = self.descs, []
        ev = torch.cuda.Event()
        ev.record()                                               # (the early reduce depends on the partial kernels issued so far ...)
        self._pre = (early, ev, fams[-1])
        return True

    def flush(self, side=None):
        """Launch what is pending and sum every slab into its parameter gradient.  side (a layers.SideStream forked from the CAPTURE'S ORIGIN
        stream - the image tower's; round 6): the pending families go smallest first with the two largest swapped, and the reduce of everything
        launched before the LAST partial kernel runs on the side stream beside that kernel (the image tower's tail: 80 % of the slab bytes
        belong to the 128-channel kernel-row launch, whose reduce - HBM-bound - then hides under the 64-channel launch - MFMA-bound)."""
        if side is None or not self.prelaunch():
            self.launch_jobs()
            self._reduce(self.descs)
            self.ci, self.off, self.descs = 0, 0, []
            self.dws.clear()
            return
        early, ev, last = self._pre
        self._pre = None
        late, self.descs = self.descs, []                         # (partial kernels issued between prelaunch() and here - the stem's)
        self.launch_jobs(last)                                    # (... and is issued behind the last one: released after it under graph replay)
        with torch.cuda.stream(side.fork(event=ev)):
            self._reduce(early)
        side.join()
        self._reduce(late + self.descs)
        self.ci, self.off, self.descs = 0, 0, []
        self.dws.clear()


def wgrad_batch(device):
    """A WgradBatch, or None when grouping is switched off (conv_wgrad then reduces per layer)."""
    return WgradBatch(device) if _WGRAD_GROUPED else None


def conv_wgrad(x, dout, g: ConvGeom, like: torch.Tensor, precision: str, row_mask=None, out_scale: float = 1.0, batch=None, rows=None):
    """Gradient of the layer's parameter, returned in the parameter's own layout (shape of ``like``), times out_scale.
    With ``batch`` (a WgradBatch of the CURRENT stream) only the partial sums are launched; batch.flush() completes dw.
    rows = (row_pos, count) from mask_compact: the contraction runs over those output positions only (pass no row_mask)."""
    assert rows is None or row_mask is None
    assert x.dtype == dout.dtype
    dw = torch.empty_like(like)
    ws = batch.slab(g.wgrad_ws) if batch is not None else _workspace(g.wgrad_ws, x.device)
    plan = g.plan(x.device)
    s_co, s_tap, s_ci = g.strides
    bi = 128 if (g.cout % 128 == 0 and g.kpad >= 128) else 64
    h16 = x.dtype != torch.float32
    bj = 128 if bi == 128 else (256 if (g.kpad <= 256 and not (h16 and g.wgrad_dma)) else 128)
    s3 = 1 if (split3(precision) and not h16) else 0          # fp32 tensors (heads, GRU) take the 3-product split in the f16 mode too
    if h16 and g.wgrad_brick and row_mask is not None:
        sym = f"conv_vox0_wgrad_kernel<{_TNAME[x.dtype]}>"
    elif h16 and g.wgrad_krow and row_mask is None and rows is None:
        sym = f"conv_wgrad_krow_kernel<{128 if g.cout % 128 == 0 else 64}, {_TNAME[x.dtype]}>"
    elif h16 and g.wgrad_dma:
        sym = f"conv_wgrad_dma_kernel<{bi}, {bj}, {_TNAME[x.dtype]}>"
    else:
        sym = f"conv_wgrad_kernel<{bi}, {bj}, {2 if s3 else 1}, {_TNAME[x.dtype]}>"
    if batch is not None and batch.group_jobs and h16 and row_mask is None:
        fam, tiles, _ = g.wgrad_group(_abf(x))
        if fam:
            xa, da = _act(x), _act(dout)
            job = _C.TriWgradJob(_C.C.pointer(g.desc), ptr(xa), ptr(da), ptr(plan), ptr(ws), ws.numel(), ptr(dw), s_co, s_tap, s_ci, g.cin,
                                 float(out_scale), ptr(rows[0]) if rows else None, ptr(rows[1]) if rows else None)
            batch.add_job(fam, tiles, job, (xa, da, ws, dw, plan, rows), _flops(g, rows), sym)
            batch.dws[dw.data_ptr()] = dw
            return dw
    if batch is not None:
        batch.dws[dw.data_ptr()] = dw
        desc = _C.TriWgradReduce()
        check(_timed(sym, _flops(g, rows, row_mask),
                     lambda: lib().tri_conv_wgrad_partial(_C.C.byref(g.desc), ptr(_act(x)), ptr(_act(dout)), ptr(row_mask), ptr(plan),
                                                          ptr(ws), ws.numel(), ptr(dw), s_co, s_tap, s_ci, g.cin, s3, _abf(x),
                                                          float(out_scale), ptr(rows[0]) if rows else None,
                                                          ptr(rows[1]) if rows else None, _C.C.byref(desc), stream())),
              "tri_conv_wgrad_partial")
        batch.descs.append(desc)
        return dw
    check(_timed(sym, _flops(g, rows, row_mask),
                 lambda: lib().tri_conv_wgrad(_C.C.byref(g.desc), ptr(_act(x)), ptr(_act(dout)), ptr(row_mask), ptr(plan), ptr(ws),
                                              ws.numel(), ptr(dw), s_co, s_tap, s_ci, g.cin, s3, _abf(x), float(out_scale),
                                              ptr(rows[0]) if rows else None, ptr(rows[1]) if rows else None, stream())),
          "tri_conv_wgrad")
    return dw


# ------------------------------------------------------------------------------------------------ BatchNorm
class BNCoeffs:
    __slots__ = ("mean", "invstd", "scale", "shift")

    def __init__(self, C, device):
        buf = torch.empty((4, C), dtype=torch.float32, device=device)
        self.mean, self.invstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]


# ---- optional synchronised BatchNorm (SURVEY 8e "BatchNorm caveat") ---------------------------------------------------------------
# Default (like the reference under Lightning DDP without sync_batchnorm): per-rank statistics.  set_sync_bn(True) makes every
# BatchNorm of the towers use GLOBAL-batch statistics - one SUM all-reduce of the per-channel (sum, sum of squares) record and of the
# row count in the forward, one of the (sum g, sum g*y) record in the backward (the parameter gradients dgamma / dbeta stay the LOCAL
# sums: the gradient all-reduce of the step adds the ranks up) - so N ranks reproduce the single-process run on the global batch
# (tests/test_gpu_dp.py::test_sync_bn_two_ranks_reproduce_the_single_process_global_batch).  25 + 25 small collectives per step: a
# parity mode, eager only (the collectives cannot sit inside a captured tower), never the benchmarked configuration.
_SYNC_BN = False


def set_sync_bn(flag: bool):
    global _SYNC_BN
    _SYNC_BN = bool(flag)


def _sync_world() -> int:
    if not _SYNC_BN:
        return 1
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _sync_sum(t: torch.Tensor) -> torch.Tensor:
    from .parallel import all_reduce_sum                  # (host-staged under gloo, RCCL under nccl)
    if t.dtype == torch.int32:                             # gloo has no int32 SUM on every build: go through int64
        u = t.to(torch.int64)
        all_reduce_sum(u)
        return u.to(torch.int32)
    all_reduce_sum(t)
    return t


def bn_finalize(stats, C, gamma, beta, running_mean, running_var, nbt, count_dev=None, count_host=0, momentum=0.1, eps=1e-5):
    world = _sync_world()
    if world > 1:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("sync_bn: the BatchNorm collectives cannot be captured into a HIP graph - run the step eagerly")
        stats = _sync_sum(stats.sum(0, keepdim=True).contiguous())
        if count_dev is not None:
            count_dev = _sync_sum(count_dev.clone())
        else:
            count_host = int(count_host) * world
    co = BNCoeffs(C, stats.device)
    check(lib().tri_bn_finalize(ptr(stats), stats.shape[0], C, ptr(count_dev), int(count_host), ptr(gamma), ptr(beta),
                                ptr(running_mean), ptr(running_var), ptr(nbt), momentum, eps, ptr(co.mean), ptr(co.invstd),
                                ptr(co.scale), ptr(co.shift), stream()), "tri_bn_finalize")
    return co


def bn_eval_coeffs(C, gamma, beta, running_mean, running_var, eps=1e-5):
    co = BNCoeffs(C, gamma.device)
    check(lib().tri_bn_eval_coeffs(C, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), eps, ptr(co.mean),
                                   ptr(co.invstd), ptr(co.scale), ptr(co.shift), stream()), "tri_bn_eval_coeffs")
    return co


def bn_act(y, co: BNCoeffs, relu=True, res=None, res_co: BNCoeffs | None = None):
    C = y.shape[-1]
    out = torch.empty_like(y)
    check(lib().tri_bn_act(ptr(_act(y)), ptr(co.scale), ptr(co.shift), ptr(res), ptr(res_co.scale) if res_co else None,
                           ptr(res_co.shift) if res_co else None, ptr(out), y.numel() // C, C, 1 if relu else 0, _abf(y), stream()),
          "tri_bn_act")
    return out


def relu_bwd(dout, out, inplace=True):
    g = dout if inplace else torch.empty_like(dout)
    check(lib().tri_relu_bwd(ptr(_act(dout)), ptr(out), ptr(g), dout.numel(), _abf(dout), stream()), "tri_relu_bwd")
    return g


def _bn_bwd_finalize(partial, nblk, C, count_dev, count_host, gamma, co: "BNCoeffs", out_scale):
    """tri_bn_bwd_finalize -> buf [5, C] = (dgamma, dbeta, c1, c2, c3).  Under sync_bn the apply coefficients come from the GLOBAL
    (all-reduced) sums and row count, dgamma / dbeta from this rank's own sums."""
    buf = torch.empty((5, C), dtype=torch.float32, device=partial.device)
    check(lib().tri_bn_bwd_finalize(ptr(partial), nblk, C, ptr(count_dev), int(count_host), ptr(gamma), ptr(co.mean),
                                    ptr(co.invstd), ptr(buf[0]), ptr(buf[1]), ptr(buf[2]), ptr(buf[3]), ptr(buf[4]), float(out_scale), stream()),
          "tri_bn_bwd_finalize")
    world = _sync_world()
    if world > 1:
        tot = _sync_sum(partial.sum(0, keepdim=True).contiguous())
        if count_dev is not None:
            count_dev = _sync_sum(count_dev.clone())
        else:
            count_host = int(count_host) * world
        gbuf = torch.empty((5, C), dtype=torch.float32, device=partial.device)
        check(lib().tri_bn_bwd_finalize(ptr(tot), 1, C, ptr(count_dev), int(count_host), ptr(gamma), ptr(co.mean), ptr(co.invstd), ptr(gbuf[0]),
                                        ptr(gbuf[1]), ptr(gbuf[2]), ptr(gbuf[3]), ptr(gbuf[4]), float(out_scale), stream()), "tri_bn_bwd_finalize")
        buf = torch.stack([buf[0], buf[1], gbuf[2], gbuf[3], gbuf[4]])
    return buf


_BN_SMALL = True                                                    # one-launch BatchNorm backward of tiny tensors (A/B switch dropped in round 6)


def bn_bwd(y, g, co: BNCoeffs, gamma, count_dev=None, count_host=0, row_mask=None, inplace=True, relu=False, relu_out=None,
           g_masked=None, out_scale: float = 1.0, keep_inactive: bool = False, partial=None):
    # row_mask: rows with 0 are never read by either pass (their y / g may be unwritten) and come out as zeros in dy
    """Returns (dy, dgamma, dbeta).  g = gradient w.r.t. the BN output (already activation-masked), or - with relu=True -
    w.r.t. relu(bn(y)): the ReLU mask is then recomputed from y inside the two passes (no separate relu_bwd pass), or -
    with relu_out - w.r.t. relu(bn(y) + residual) whose saved output is relu_out; g_masked (may alias g) then receives
    g * (relu_out > 0), the gradient of the pre-activation sum that the residual branch needs too.
    out_scale multiplies dgamma / dbeta only (f16 mode: g and dy carry F16_GRAD_SCALE, parameter gradients do not)."""
    rs, rb = (co.scale, co.shift) if relu else (None, None)
    C = y.shape[-1]
    M = y.numel() // C
    assert y.dtype == g.dtype
    if partial is None and _BN_SMALL and M <= 512 and C % 8 == 0 and C >= 64 and _sync_world() == 1:
        # tiny tensors (deepest voxel level): sums, coefficients and the apply pass in ONE launch (tri_bn_bwd_small; slower than the
        # three passes from ~1 k rows on - its loads are 16 bytes per cache line)
        dy = g if inplace else torch.empty_like(g)
        buf = torch.empty((2, C), dtype=torch.float32, device=y.device)
        check(lib().tri_bn_bwd_small(ptr(_act(y)), ptr(_act(g)), M, C, ptr(count_dev), int(count_host), ptr(gamma), ptr(co.mean), ptr(co.invstd),
                                     ptr(rs), ptr(rb), ptr(relu_out), ptr(g_masked), ptr(row_mask), 1 if keep_inactive else 0, ptr(dy), ptr(buf[0]),
                                     ptr(buf[1]), float(out_scale), _abf(y), stream()), "tri_bn_bwd_small")
        return dy, buf[0], buf[1]
    if partial is not None:                          # the sums came out of the data gradient that produced g (conv_dgrad(bn_sums=...))
        assert row_mask is None and partial.shape[1:] == (2, C)
        nblk = partial.shape[0]
    else:
        nblk = lib().tri_bn_bwd_num_blocks(M)
        partial = torch.empty((nblk, 2, C), dtype=torch.float32, device=y.device)
        check(lib().tri_bn_bwd_reduce(ptr(_act(y)), ptr(_act(g)), M, C, ptr(partial), ptr(rs), ptr(rb), ptr(relu_out), ptr(row_mask), _abf(y), stream()),
              "tri_bn_bwd_reduce")
    buf = _bn_bwd_finalize(partial, nblk, C, count_dev, count_host, gamma, co, out_scale)
    dy = g if inplace else torch.empty_like(g)
    check(lib().tri_bn_bwd_apply(ptr(y), ptr(g), ptr(buf[2]), ptr(buf[3]), ptr(buf[4]), ptr(row_mask), ptr(dy), M, C, ptr(rs), ptr(rb),
                                 ptr(relu_out), ptr(g_masked), 1 if keep_inactive else 0, _abf(y), stream()), "tri_bn_bwd_apply")
    return dy, buf[0], buf[1]


def bn_bwd_pair(ya, coa: BNCoeffs, gamma_a, yb, cob: BNCoeffs, gamma_b, g, relu_out, count_host, g_masked=None, out_scale: float = 1.0):
    """BatchNorm backward of bn_a(ya) and bn_b(yb) under out = relu(bn_a(ya) + bn_b(yb)) (bn2 and the shortcut's BatchNorm of a
    down-sampling BasicBlock): -> (dya, dgamma_a, dbeta_a, dyb, dgamma_b, dbeta_b); g_masked (may alias g) receives g * (out > 0).
    Three launches for both tensors; the values of bn_bwd(ya, g, ..., relu_out=out, g_masked=...) followed by bn_bwd(yb, g_masked, ...)."""
    C = ya.shape[-1]
    M = ya.numel() // C
    assert ya.dtype == g.dtype == yb.dtype == relu_out.dtype and ya.shape == yb.shape and _sync_world() == 1
    nblk = lib().tri_bn_bwd_num_blocks(M)
    partial = torch.empty((2, nblk, 2, C), dtype=torch.float32, device=ya.device)
    check(lib().tri_bn_bwd_pair_reduce(ptr(_act(ya)), ptr(_act(yb)), ptr(_act(g)), ptr(relu_out), M, C, ptr(partial[0]), ptr(partial[1]), _abf(ya),
                                       stream()), "tri_bn_bwd_pair_reduce")
    buf = torch.empty((2, 5, C), dtype=torch.float32, device=ya.device)
    check(lib().tri_bn_bwd_pair_finalize(ptr(partial[0]), ptr(partial[1]), nblk, C, int(count_host), ptr(gamma_a), ptr(coa.mean), ptr(coa.invstd),
                                         ptr(buf[0]), ptr(gamma_b), ptr(cob.mean), ptr(cob.invstd), ptr(buf[1]), float(out_scale), stream()),
          "tri_bn_bwd_pair_finalize")
    dya, dyb = torch.empty_like(ya), torch.empty_like(yb)
    check(lib().tri_bn_bwd_pair_apply(ptr(ya), ptr(yb), ptr(g), ptr(relu_out), ptr(buf[0]), ptr(buf[1]), ptr(dya), ptr(dyb), ptr(g_masked), M, C,
                                      _abf(ya), stream()), "tri_bn_bwd_pair_apply")
    return dya, buf[0, 0], buf[0, 1], dyb, buf[1, 0], buf[1, 1]


# ------------------------------------------------------------------------------------------------ pooling
def bn_relu_pool3d_fwd(y, co: BNCoeffs, mask, B, D, C, want_mask: bool = True):
    """want_mask=False: the pooled level's site mask is not written (the caller has it from mask_pyramid); returns (pooled, None)."""
    Do = D // 2
    pooled = torch.empty((B, Do, Do, Do, C), dtype=y.dtype, device=y.device)
    mask_out = torch.empty(((B * Do ** 3 + 31) // 32 * 32,), dtype=torch.uint8, device=y.device) if want_mask else None   # every byte written by the kernel
    check(lib().tri_bn_relu_pool3d_fwd(ptr(_act(y)), ptr(co.scale), ptr(co.shift), ptr(mask), B, D, C, ptr(pooled), ptr(mask_out),
                                       _abf(y), stream()), "tri_bn_relu_pool3d_fwd")
    return pooled, mask_out


def pool3d_bwd_route(y, co: BNCoeffs, mask, pooled, dpooled, B, D, C):
    g = torch.empty_like(y)
    assert dpooled.dtype == y.dtype and pooled.dtype == y.dtype
    check(lib().tri_pool3d_bwd_route(ptr(_act(y)), ptr(co.scale), ptr(co.shift), ptr(mask), ptr(pooled), ptr(_act(dpooled)), B, D, C,
                                     ptr(g), _abf(y), stream()), "tri_pool3d_bwd_route")
    return g


_VOX_ROWS = True                                                    # row-list passes on the fine voxel levels (A/B switch dropped in round 6)
_ROUTE_RED = True                                                     # BatchNorm-backward sums inside the row-list routing walk (A/B switch dropped in round 6)


def pool3d_bwd_route_rows(y, co: BNCoeffs, mask, pooled, dpooled, B, D, C, rows_out):
    """pool3d_bwd_route over the ACTIVE pooled sites only (rows_out = the next level's (row_pos, count))."""
    g = torch.empty_like(y)
    check(lib().tri_pool3d_bwd_route_rows(ptr(_act(y)), ptr(co.scale), ptr(co.shift), ptr(mask), ptr(pooled), ptr(_act(dpooled)), B, D, C,
                                          ptr(g), ptr(rows_out[0]), ptr(rows_out[1]), _abf(y), stream()), "tri_pool3d_bwd_route_rows")
    return g


def bn_bwd_rows(y, g, co: BNCoeffs, gamma, rows, out_scale: float = 1.0):
    """BatchNorm backward over a compact row list, in place on g: rows outside the list are neither read nor written."""
    C = y.shape[-1]
    scratch = torch.empty((lib().tri_bn_bwd_rows_scratch(C),), dtype=torch.uint8, device=y.device)
    buf = torch.empty((2, C), dtype=torch.float32, device=y.device)
    check(lib().tri_bn_bwd_rows(ptr(_act(y)), ptr(_act(g)), C, ptr(rows[0]), ptr(rows[1]), rows[0].numel(), ptr(gamma), ptr(co.mean),
                                ptr(co.invstd), ptr(g), ptr(buf[0]), ptr(buf[1]), float(out_scale), ptr(scratch), _abf(y), stream()),
          "tri_bn_bwd_rows")
    return g, buf[0], buf[1]


def pool3d_bn_bwd(y, co: BNCoeffs, mask, pooled, dpooled, B, D, C, gamma, count_dev, out_scale: float = 1.0, fused: bool = True,
                  keep_inactive: bool = False, rows=None, rows_out=None):
    """Backward of BatchNorm1d -> ReLU -> SparseMaxPool3d of one voxel level: (dy, dgamma, dbeta).  fused: the routing pass also
    produces the BatchNorm-backward sums (tri_pool3d_bwd_route_reduce), so the level costs route + finalize + apply instead of
    route + reduce + finalize + apply; channel counts whose quads do not divide 256 take the unfused passes.  keep_inactive: dy rows
    of inactive sites are left unwritten instead of zeroed (level 0: its dy only feeds the weight gradient over the same mask - at
    13 % occupancy the zeros were 87 % of the pass's writes)."""
    # rows / rows_out (round 4): (row_pos, count) of THIS level / of the pooled level.  The finest levels are 13-16 % occupied: the
    # routing pass then walks the active pooled sites, and - where dy may stay unwritten at inactive sites (keep_inactive) - the whole
    # BatchNorm backward walks this level's list
    big = y.numel() // C > 16384
    if _VOX_ROWS and _ROUTE_RED and fused and big and C % 4 == 0 and 256 % (C // 4) == 0 and rows_out is not None:
        # (round 5) the routing walk over the active pooled sites also leaves the BatchNorm-backward sums: route + finalize + apply
        # (fused=False / TRICOLO_POOL_REDUCE=0 select the separate route and reduce passes here too)
        assert dpooled.dtype == y.dtype and pooled.dtype == y.dtype
        g = torch.empty_like(y)
        nblk = lib().tri_pool3d_bwd_route_rows_num_blocks(B, D, C)
        partial = torch.empty((nblk, 2, C), dtype=torch.float32, device=y.device)
        check(lib().tri_pool3d_bwd_route_rows_reduce(ptr(_act(y)), ptr(co.scale), ptr(co.shift), ptr(mask), ptr(pooled), ptr(_act(dpooled)),
                                                     B, D, C, ptr(g), ptr(rows_out[0]), ptr(rows_out[1]), ptr(partial), _abf(y), stream()),
              "tri_pool3d_bwd_route_rows_reduce")
        buf = _bn_bwd_finalize(partial, nblk, C, count_dev, 0, gamma, co, out_scale)
        if rows is not None and keep_inactive:
            check(lib().tri_bn_bwd_apply_rows(ptr(y), ptr(g), ptr(buf[2]), ptr(buf[3]), ptr(buf[4]), ptr(g), C, ptr(rows[0]), ptr(rows[1]),
                                              rows[0].numel(), _abf(y), stream()), "tri_bn_bwd_apply_rows")
        else:
            check(lib().tri_bn_bwd_apply(ptr(y), ptr(g), ptr(buf[2]), ptr(buf[3]), ptr(buf[4]), ptr(mask), ptr(g), y.numel() // C, C,
                                         None, None, None, None, 1 if keep_inactive else 0, _abf(y), stream()), "tri_bn_bwd_apply")
        return g, buf[0], buf[1]
    if _VOX_ROWS and big and _sync_world() == 1 and C % 4 == 0 and (rows_out is not None or (rows is not None and keep_inactive)):
        gz = (pool3d_bwd_route_rows(y, co, mask, pooled, dpooled, B, D, C, rows_out) if rows_out is not None
              else pool3d_bwd_route(y, co, mask, pooled, dpooled, B, D, C))
        if rows is not None and keep_inactive:
            return bn_bwd_rows(y, gz, co, gamma, rows, out_scale=out_scale)
        return bn_bwd(y, gz, co, gamma, count_dev=count_dev, row_mask=mask, out_scale=out_scale, keep_inactive=keep_inactive)
    if C % 4 or (C // 4) > 256 or 256 % (C // 4) or not fused:
        gz = pool3d_bwd_route(y, co, mask, pooled, dpooled, B, D, C)
        return bn_bwd(y, gz, co, gamma, count_dev=count_dev, row_mask=mask, out_scale=out_scale, keep_inactive=keep_inactive)
    assert dpooled.dtype == y.dtype and pooled.dtype == y.dtype
    g = torch.empty_like(y)
    nblk = lib().tri_pool3d_bwd_route_reduce_num_blocks(B, D, C)
    partial = torch.empty((nblk, 2, C), dtype=torch.float32, device=y.device)
    check(lib().tri_pool3d_bwd_route_reduce(ptr(_act(y)), ptr(co.scale), ptr(co.shift), ptr(mask), ptr(pooled), ptr(_act(dpooled)), B, D, C,
                                            ptr(g), ptr(partial), _abf(y), stream()), "tri_pool3d_bwd_route_reduce")
    M = y.numel() // C
    buf = _bn_bwd_finalize(partial, nblk, C, count_dev, 0, gamma, co, out_scale)
    check(lib().tri_bn_bwd_apply(ptr(y), ptr(g), ptr(buf[2]), ptr(buf[3]), ptr(buf[4]), ptr(mask), ptr(g), M, C, None, None, None, None,
                                 1 if keep_inactive else 0, _abf(y), stream()), "tri_bn_bwd_apply")
    return g, buf[0], buf[1]


def maxpool2d_fwd(x, want_arg=True, bn: BNCoeffs = None):
    """3x3/2/pad-1 max-pool; also returns the winning-tap byte map used by maxpool2d_bwd.
    With ``bn`` the pooled tensor is relu(x * scale + shift): BatchNorm + ReLU + MaxPool2d in one pass."""
    N, _, H, W, C = x.shape
    out = torch.empty((N, 1, (H + 1) // 2, (W + 1) // 2, C), dtype=x.dtype, device=x.device)
    arg = torch.empty(out.shape, dtype=torch.uint8, device=x.device) if want_arg else None
    check(lib().tri_maxpool2d_fwd(ptr(_act(x)), N, H, W, C, ptr(out), ptr(arg), ptr(bn.scale if bn else None), ptr(bn.shift if bn else None),
                                  _abf(x), stream()), "tri_maxpool2d_fwd")
    return out, arg


_STEM_POOLED = True                                                    # stem BatchNorm-backward sums from the pooled tensors (A/B switch dropped in round 6)


def _maxpool_bn_bwd_sums(y, arg, dpool, co: "BNCoeffs", gamma, pooled):
    """-> (partial [blocks, 2, C], blocks): sums of g and g * y over the stem's conv output, g = gradient of relu(bn(y)) in front of
    the max-pool.  With the forward's pooled output at hand (16-bit storage) the sums are taken over the pooled-resolution tensors."""
    N, _, H, W, C = y.shape
    if pooled is not None and _STEM_POOLED and y.dtype != torch.float32 and C % 8 == 0 and 256 % (C // 8) == 0:
        assert pooled.dtype == y.dtype and pooled.numel() == dpool.numel()
        nblk = lib().tri_maxpool_bn_bwd_pooled_num_blocks(N, H, W)
        partial = torch.empty((nblk, 2, C), dtype=torch.float32, device=y.device)
        check(lib().tri_maxpool_bn_bwd_reduce_pooled(ptr(_act(pooled)), ptr(_act(dpool)), ptr(_act(y)), ptr(arg), N, H, W, C, ptr(partial),
                                                     ptr(co.scale), ptr(co.shift), ptr(_f32(gamma.detach())), _abf(y), stream()),
              "tri_maxpool_bn_bwd_reduce_pooled")
        return partial, nblk
    nblk = lib().tri_maxpool_bn_bwd_num_blocks(N, H, W)
    partial = torch.empty((nblk, 2, C), dtype=torch.float32, device=y.device)
    check(lib().tri_maxpool_bn_bwd_reduce(ptr(_act(y)), ptr(arg), ptr(_act(dpool)), N, H, W, C, ptr(partial), ptr(co.scale), ptr(co.shift),
                                          _abf(y), stream()), "tri_maxpool_bn_bwd_reduce")
    return partial, nblk


def maxpool_bn_bwd(y, arg, dpool, co: "BNCoeffs", gamma, out_scale: float = 1.0, pooled=None):
    """Backward of relu(bn(y)) -> MaxPool2d(3, 2, 1) in the two BatchNorm passes (no max-pool backward pass, no 4x-sized gradient
    tensor): (dy, dgamma, dbeta).  y [N,1,H,W,C], arg / dpool [N,1,H/2,W/2,C]; H, W even.  pooled: the forward's max-pool output."""
    N, _, H, W, C = y.shape
    assert dpool.dtype == y.dtype and H % 2 == 0 and W % 2 == 0
    partial, nblk = _maxpool_bn_bwd_sums(y, arg, dpool, co, gamma, pooled)
    buf = _bn_bwd_finalize(partial, nblk, C, None, N * H * W, gamma, co, out_scale)
    dy = torch.empty_like(y)
    check(lib().tri_maxpool_bn_bwd_apply(ptr(y), ptr(arg), ptr(dpool), N, H, W, C, ptr(buf[2]), ptr(buf[3]), ptr(buf[4]), ptr(co.scale),
                                         ptr(co.shift), ptr(dy), _abf(y), stream()), "tri_maxpool_bn_bwd_apply")
    return dy, buf[0], buf[1]


def maxpool_bn_bwd_wgrad(x0, y, arg, dpool, co: "BNCoeffs", gamma, g: "ConvGeom", like, precision: str, out_scale: float = 1.0, batch=None,
                         pooled=None):
    """Stem backward in three launches: (dw, dgamma, dbeta) of conv -> BN -> ReLU -> MaxPool2d(3, 2, 1) (mv_cnn.py:44).  As maxpool_bn_bwd,
    but the apply pass is folded into the weight-gradient kernel's operand staging (tri_conv_stem_wgrad_bn): the stem has no data
    gradient, so the gradient w.r.t. the conv output - 100 MB written and re-read at the bench shape - is never stored.  Falls back to
    maxpool_bn_bwd + conv_wgrad where the layer does not run the stem kernel."""
    N, _, H, W, C = y.shape
    h16 = y.dtype != torch.float32
    if not (h16 and C == 64 and H % 2 == 0 and W % 2 == 0 and dpool.dtype == y.dtype and _sync_world() == 1):
        dy, dgamma, dbeta = maxpool_bn_bwd(y, arg, dpool, co, gamma, out_scale=out_scale, pooled=pooled)
        return conv_wgrad(x0, dy, g, like, precision, out_scale=out_scale, batch=batch), dgamma, dbeta
    partial, nblk = _maxpool_bn_bwd_sums(y, arg, dpool, co, gamma, pooled)
    buf = _bn_bwd_finalize(partial, nblk, C, None, N * H * W, gamma, co, out_scale)
    dw = torch.empty_like(like)
    mark = (batch.ci, batch.off) if batch is not None else None   # the arena cursor: an unsupported geometry hands its slab back
    ws = batch.slab(g.wgrad_ws) if batch is not None else _workspace(g.wgrad_ws, y.device)
    s_co, s_tap, s_ci = g.strides
    desc = _C.TriWgradReduce()
    rc = _timed(f"conv_stem_wgrad_kernel<{g.kernel[1]}, {_TNAME[y.dtype]}>", g.flops,
                lambda: lib().tri_conv_stem_wgrad_bn(_C.C.byref(g.desc), ptr(_act(x0)), ptr(_act(y)), ptr(arg), ptr(_act(dpool)), ptr(buf[2]),
                                                     ptr(buf[3]), ptr(buf[4]), ptr(co.scale), ptr(co.shift), ptr(ws), ws.numel(), ptr(dw), s_co,
                                                     s_tap, s_ci, g.cin, _abf(y), float(out_scale), _C.C.byref(desc), stream()))
    if rc == _C.TRI_ERR_UNSUPPORTED:                               # not a stem-kernel geometry: the two-pass form
        if batch is not None:
            batch.ci, batch.off = mark                             # (nothing was launched into that slab: conv_wgrad below carves its own)
        dy = torch.empty_like(y)
        check(lib().tri_maxpool_bn_bwd_apply(ptr(y), ptr(arg), ptr(dpool), N, H, W, C, ptr(buf[2]), ptr(buf[3]), ptr(buf[4]), ptr(co.scale),
                                             ptr(co.shift), ptr(dy), _abf(y), stream()), "tri_maxpool_bn_bwd_apply")
        return conv_wgrad(x0, dy, g, like, precision, out_scale=out_scale, batch=batch), buf[0], buf[1]
    check(rc, "tri_conv_stem_wgrad_bn")
    if batch is not None:
        batch.descs.append(desc)
        batch.dws[dw.data_ptr()] = dw
    else:
        check(lib().tri_wgrad_reduce_grouped(_C.C.byref(desc), 1, stream()), "tri_wgrad_reduce_grouped")
    return dw, buf[0], buf[1]


def maxpool2d_bwd(arg, dout, in_shape):
    N, _, H, W, C = in_shape
    dx = torch.empty(in_shape, dtype=dout.dtype, device=dout.device)
    check(lib().tri_maxpool2d_bwd(ptr(arg), ptr(_act(dout)), N, H, W, C, ptr(dx), _abf(dout), stream()), "tri_maxpool2d_bwd")
    return dx


def avgpool_viewmax_fwd(x, B, V):
    N, _, H, W, C = x.shape
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    arg = torch.empty((B, C), dtype=torch.int32, device=x.device)
    check(lib().tri_avgpool_viewmax_fwd(ptr(_act(x)), B, V, H * W, C, ptr(out), ptr(arg), _abf(x), stream()), "tri_avgpool_viewmax_fwd")
    return out, arg


def avgpool_viewmax_bwd(dout, arg, shape, B, V, dtype=torch.float32, scale: float = 1.0):
    N, _, H, W, C = shape
    dx = torch.empty(shape, dtype=dtype, device=dout.device)
    check(lib().tri_avgpool_viewmax_bwd(ptr(_f32(dout)), ptr(arg), B, V, H * W, C, ptr(dx), _abf(dx), float(scale), stream()),
          "tri_avgpool_viewmax_bwd")
    return dx


def cast_from_f32(x, dtype, scale: float = 1.0):
    """scale * x (fp32) stored as `dtype` - the fp32 head -> 16-bit tower boundary (one launch, no ATen copy)."""
    if dtype == torch.float32 and scale == 1.0:
        return x
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib().tri_cast_from_f32(ptr(_f32(x)), ptr(out), x.numel(), float(scale), _abf(out), stream()), "tri_cast_from_f32")
    return out


def cast_to_f32(x):
    if x.dtype == torch.float32:
        return x
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().tri_cast_to_f32(ptr(_act(x)), ptr(out), x.numel(), _abf(x), stream()), "tri_cast_to_f32")
    return out


# ------------------------------------------------------------------------------------------------ layouts
def voxel_scatter(locs, feats, B, V, dtype=torch.float32):
    n = locs.shape[0]
    sites = B * V ** 3
    esz = torch.empty((), dtype=dtype).element_size()
    buf = torch.empty((sites * 4 * esz + (sites + 31) // 32 * 32,), dtype=torch.uint8, device=feats.device)   # grid + mask: one zero-fill
    dense = buf[:sites * 4 * esz].view(dtype).view(B, V, V, V, 4)
    mask = buf[sites * 4 * esz:]
    locs = locs.to(torch.int32).contiguous()
    check(lib().tri_voxel_scatter(ptr(locs), ptr(_f32(feats.contiguous())), n, B, V, ptr(dense), ptr(mask), _abf(dense), stream()),
          "tri_voxel_scatter")
    return dense, mask


def voxel_from_rgba(rgba_u8, dtype=torch.float32):
    """Dense RGBA u8 grids [B,4,V,V,V] (the dataset's storage format) -> (channels-last grid [B,V,V,V,4], site mask)."""
    B, C, V = rgba_u8.shape[0], rgba_u8.shape[1], rgba_u8.shape[2]
    assert C == 4 and rgba_u8.dtype == torch.uint8 and rgba_u8.shape[2:] == (V, V, V)
    dense = torch.empty((B, V, V, V, 4), dtype=dtype, device=rgba_u8.device)
    sites = B * V ** 3
    mask = torch.empty(((sites + 31) // 32 * 32,), dtype=torch.uint8, device=rgba_u8.device)      # fully written by the call
    check(lib().tri_voxel_from_rgba_u8(ptr(rgba_u8.contiguous()), B, V, ptr(dense), ptr(mask), _abf(dense), stream()), "tri_voxel_from_rgba_u8")
    return dense, mask


def mask_compact(mask, n):
    """(row_pos [n] int32 - the first `count` entries are the active positions in ascending order, count [1] int32 on the device)."""
    row_pos = torch.empty((n,), dtype=torch.int32, device=mask.device)
    count = torch.empty((1,), dtype=torch.int32, device=mask.device)
    scratch = torch.empty((lib().tri_mask_compact_scratch(n),), dtype=torch.uint8, device=mask.device)
    check(lib().tri_mask_compact(ptr(mask), n, ptr(row_pos), ptr(count), ptr(scratch), stream()), "tri_mask_compact")
    return row_pos, count


def mask_pyramid(mask0, B, V):
    """Site masks of voxel levels 1-4 (2x2x2 OR-pool of the level below) from the level-0 mask, one launch; V % 16 == 0."""
    import ctypes
    outs = [torch.empty(((B * (V >> l) ** 3 + 31) // 32 * 32,), dtype=torch.uint8, device=mask0.device) for l in range(1, 5)]
    arr = (ctypes.c_void_p * 4)(*[o.data_ptr() for o in outs])
    check(lib().tri_mask_pyramid(ptr(mask0), B, V, arr, stream()), "tri_mask_pyramid")
    return outs


MASK_MULTI_MAX = 1024 * 2048          # sites per list tri_mask_compact_multi takes (1,024 blocks: the scan is folded into the write pass)


def mask_compact_multi(masks, ns):
    """mask_compact of several masks in two launches: [(row_pos, count), ...].  Lists beyond MASK_MULTI_MAX sites go through mask_compact."""
    import ctypes
    out = [None] * len(masks)
    small = [i for i, n in enumerate(ns) if n <= MASK_MULTI_MAX]
    for i, n in enumerate(ns):
        if n > MASK_MULTI_MAX:
            out[i] = mask_compact(masks[i], n)
    for s0 in range(0, len(small), 8):
        grp = small[s0:s0 + 8]
        dev = masks[grp[0]].device
        rows = [torch.empty((ns[i],), dtype=torch.int32, device=dev) for i in grp]
        counts = torch.empty((len(grp),), dtype=torch.int32, device=dev)
        n_arr = (ctypes.c_long * len(grp))(*[ns[i] for i in grp])
        scratch = torch.empty((lib().tri_mask_compact_multi_scratch(n_arr, len(grp)),), dtype=torch.uint8, device=dev)
        m_arr = (ctypes.c_void_p * len(grp))(*[masks[i].data_ptr() for i in grp])
        r_arr = (ctypes.c_void_p * len(grp))(*[r.data_ptr() for r in rows])
        c_arr = (ctypes.c_void_p * len(grp))(*[counts[k:k + 1].data_ptr() for k in range(len(grp))])
        check(lib().tri_mask_compact_multi(m_arr, n_arr, len(grp), r_arr, c_arr, ptr(scratch), stream()), "tri_mask_compact_multi")
        for k, i in enumerate(grp):
            out[i] = (rows[k], counts[k:k + 1])
    return out


def mask_count(mask, n):
    cnt = torch.empty((1,), dtype=torch.int32, device=mask.device)
    check(lib().tri_mask_count(ptr(mask), n, ptr(cnt), stream()), "tri_mask_count")
    return cnt


def nchw3_to_nhwc4(x, dtype=torch.float32):
    N, C, H, W = x.shape
    assert C == 3
    out = torch.empty((N, 1, H, W, 4), dtype=dtype, device=x.device)
    check(lib().tri_nchw3_to_nhwc4(ptr(_f32(x.contiguous())), N, H, W, ptr(out), _abf(out), stream()), "tri_nchw3_to_nhwc4")
    return out


_CLIP_MEAN = (C_F3 := _C.C.c_float * 3)(0.48145466, 0.4578275, 0.40821073)      # general_dataset.py:88
_CLIP_STD = C_F3(0.26862954, 0.26130258, 0.27577711)


def nchw3_u8_to_nhwc4(x_u8, dtype=torch.float32, mean=None, std=None):
    """u8 images [N,3,H,W] -> channels-last [N,1,H,W,4], ToTensor + CLIP Normalize done on the device."""
    N, C, H, W = x_u8.shape
    assert C == 3 and x_u8.dtype == torch.uint8
    out = torch.empty((N, 1, H, W, 4), dtype=dtype, device=x_u8.device)
    m = C_F3(*mean) if mean is not None else _CLIP_MEAN
    sd = C_F3(*std) if std is not None else _CLIP_STD
    check(lib().tri_nchw3_u8_to_nhwc4(ptr(x_u8.contiguous()), N, H, W, _C.C.cast(m, _C.C.c_void_p), _C.C.cast(sd, _C.C.c_void_p), ptr(out),
                                      _abf(out), stream()), "tri_nchw3_u8_to_nhwc4")
    return out


# ------------------------------------------------------------------------------------------------ row ops
def l2norm_fwd(x, eps=1e-12):
    rows, D = x.shape
    z = torch.empty_like(x)
    norm = torch.empty((rows,), dtype=torch.float32, device=x.device)
    check(lib().tri_l2norm_fwd(ptr(_f32(x)), rows, D, eps, ptr(z), ptr(norm), stream()), "tri_l2norm_fwd")
    return z, norm


def l2norm_bwd(z, norm, dz, eps=1e-12):
    rows, D = z.shape
    dx = torch.empty_like(z)
    check(lib().tri_l2norm_bwd(ptr(z), ptr(norm), ptr(_f32(dz.contiguous())), rows, D, eps, ptr(dx), stream()), "tri_l2norm_bwd")
    return dx


# ------------------------------------------------------------------------------------------------ token embedding
def embedding_fwd(tokens, weight):
    """tokens [B,L] int32, weight [V,D] -> [L,B,D] (time-major, what the GRU consumes)."""
    B, L = tokens.shape
    D = weight.shape[1]
    out = torch.empty((L, B, D), dtype=torch.float32, device=weight.device)
    check(lib().tri_embedding_fwd(ptr(tokens), ptr(_f32(weight)), B, L, D, ptr(out), stream()), "tri_embedding_fwd")
    return out


def embedding_bwd(tokens, dout, V, padding_idx=0):
    B, L = tokens.shape
    D = dout.shape[-1]
    dw = torch.empty((V, D), dtype=torch.float32, device=dout.device)
    check(lib().tri_embedding_bwd(ptr(tokens), ptr(_f32(dout.contiguous())), B, L, V, D, padding_idx, ptr(dw), stream()), "tri_embedding_bwd")
    return dw


# ------------------------------------------------------------------------------------------------ retrieval
def retrieval_topk(text, shape, labels, k=5):
    """text [Nq,D], shape [Ns,D] fp32 on the GPU, labels [Nq] int32 -> (indices [Nq,k] i32, sims [Nq,k] f64, first_hit [Nq] i32)."""
    Nq, D = text.shape
    Ns = shape.shape[0]
    idx = torch.empty((Nq, k), dtype=torch.int32, device=text.device)
    sim = torch.empty((Nq, k), dtype=torch.float64, device=text.device)
    hit = torch.empty((Nq,), dtype=torch.int32, device=text.device)
    check(lib().tri_retrieval_topk(ptr(_f32(text)), ptr(_f32(shape)), ptr(labels), Nq, Ns, D, k, ptr(idx), ptr(sim), ptr(hit), stream()),
          "tri_retrieval_topk")
    return idx, sim, hit


# ------------------------------------------------------------------------------------------------ small dense layers
def linear_small_supported(rows, K, N):
    return bool(lib().tri_linear_small_supported(int(rows), int(K), int(N)))


def linear_small_fwd(x, w, b, act, precision):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    check(_timed("linear_small_fwd_kernel", 2.0 * M * K * N,
                 lambda: lib().tri_linear_small_fwd(ptr(_f32(x)), ptr(_f32(w)), ptr(b), ptr(y), M, K, N, act, split3(precision),
                                                    stream())), "tri_linear_small_fwd")
    return y


def linear_small_bwd(x, w, out, dout, act, precision, need_dx=True, need_db=True):
    M, K = x.shape
    N = w.shape[0]
    s3 = split3(precision)
    dout = _f32(dout.contiguous())
    dw = torch.empty_like(w)
    db = torch.empty((N,), dtype=torch.float32, device=x.device) if need_db else None
    if need_dx:                                              # weight- and data-gradient tiles in one launch
        dx = torch.empty((M, K), dtype=torch.float32, device=x.device)
        check(lib().tri_linear_small_bwd(ptr(_f32(x)), ptr(dout), ptr(out), ptr(_f32(w)), ptr(dx), ptr(dw), ptr(db), M, K, N, act, s3,
                                         stream()), "tri_linear_small_bwd")
        return dx, dw, db
    check(lib().tri_linear_small_wgrad(ptr(_f32(x)), ptr(dout), ptr(out), ptr(dw), ptr(db), M, K, N, act, s3, stream()), "tri_linear_small_wgrad")
    return None, dw, db


def colsum(g):
    C = g.shape[-1]
    out = torch.empty((C,), dtype=torch.float32, device=g.device)
    check(lib().tri_colsum(ptr(_f32(g)), g.numel() // C, C, ptr(out), stream()), "tri_colsum")
    return out


def act_bwd(dout, out, act, inplace=True):
    g = dout if inplace else torch.empty_like(dout)
    check(lib().tri_act_bwd(ptr(dout), ptr(out), ptr(g), dout.numel(), act, stream()), "tri_act_bwd")
    return g


# ------------------------------------------------------------------------------------------------ GRU recurrence
def gru_mode(precision: str) -> int:
    """Operand mode of the GRU recurrence (tri_gru_fwd / tri_gru_bwd `split3`): 1 = 3-product bf16 split (bf16x3), 0 = single bf16
    products (bf16), 2 = single f16 products (f16 mode: z within 4e-5 of float64 over the 96 steps, a third of the split's MFMAs;
    TRICOLO_GRU_F16=0 keeps the split there - the A/B partner)."""
    if precision == "f16" and os.environ.get("TRICOLO_GRU_F16", "1") != "0":
        return 2
    return split3(precision)


def gru_fwd(xproj, w_hh, b_hh, B, L, precision):
    dev = xproj.device
    hs = torch.empty((2, L, B, 128), dtype=torch.float32, device=dev)
    gates = torch.empty((2, L, B, 128, 4), dtype=torch.float32, device=dev)        # (r, z, n, hn) per unit: one 16-byte access
    hfinal = torch.empty((B, 256), dtype=torch.float32, device=dev)
    check(lib().tri_gru_fwd(ptr(_f32(xproj)), ptr(_f32(w_hh)), ptr(_f32(b_hh)), B, L, ptr(hs), ptr(gates), ptr(hfinal),
                            gru_mode(precision), stream()), "tri_gru_fwd")
    return hfinal, hs, gates


def gru_bwd(dhfinal, w_hh, hs, gates, B, L, precision):
    dev = hs.device
    dgi = torch.empty((L * B, 768), dtype=torch.float32, device=dev)
    dgh = torch.empty((2, L * B, 384), dtype=torch.float32, device=dev)
    hprev = torch.empty((2, L * B, 128), dtype=torch.float32, device=dev)
    dbias = torch.empty(((B + 15) // 16, 2, 4, 128), dtype=torch.float32, device=dev)
    check(lib().tri_gru_bwd(ptr(_f32(dhfinal.contiguous())), ptr(_f32(w_hh)), ptr(hs), ptr(gates), B, L, ptr(dgi), ptr(dgh), ptr(hprev),
                            ptr(dbias), gru_mode(precision), stream()), "tri_gru_bwd")
    return dgi, dgh, hprev, dbias


# ------------------------------------------------------------------------------------------------ NT-Xent
def copy_segments(pairs):
    """[(src, dst), ...] (<= 8 contiguous fp32 tensors each way, equal sizes per pair) copied in ONE launch."""
    n = len(pairs)
    PA, LA = _C.C.c_void_p * n, _C.C.c_long * n
    for s, d in pairs:
        assert s.numel() == d.numel() and s.is_contiguous() and d.is_contiguous() and s.dtype == d.dtype == torch.float32
    check(lib().tri_copy_segments(PA(*[ptr(s) for s, _ in pairs]), PA(*[ptr(d) for _, d in pairs]), LA(*[s.numel() for s, _ in pairs]), n,
                                  stream()), "tri_copy_segments")


def gru_bias_grads(dbias):
    """dbias [nchunk, 2, 4, 128] from gru_bwd -> (db_ih_f, db_hh_f, db_ih_r, db_hh_r), [384] each, one launch."""
    out = torch.empty((4, 384), dtype=torch.float32, device=dbias.device)
    check(lib().tri_gru_bias_grads(ptr(_f32(dbias)), dbias.shape[0], ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), stream()),
          "tri_gru_bias_grads")
    return out[0], out[1], out[2], out[3]


def ntxent_fwd(za, zb, temperature, alpha, norm=True):
    """Loss only; returns (loss, workspace) - hand the workspace to ntxent_bwd."""
    B, D = za.shape
    loss = torch.empty((), dtype=torch.float32, device=za.device)
    nbytes = lib().tri_ntxent_workspace(B, D)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=za.device)
    check(lib().tri_ntxent_fwd_bwd(ptr(_f32(za)), ptr(_f32(zb)), B, D, float(temperature), float(alpha), 1 if norm else 0, ptr(loss), None,
                                   None, ptr(ws), nbytes, stream()), "tri_ntxent_fwd_bwd")
    return loss, ws


def ntxent_bwd(za, zb, ws, temperature, alpha, norm=True, dloss=None):
    """(dza, dzb) times the upstream scalar dloss (device tensor) in one launch."""
    B, D = za.shape
    dza, dzb = torch.empty_like(za), torch.empty_like(zb)
    check(lib().tri_ntxent_bwd(ptr(_f32(za)), ptr(_f32(zb)), B, D, float(temperature), float(alpha), 1 if norm else 0, ptr(dloss), ptr(dza),
                               ptr(dzb), ptr(ws), ws.numel(), stream()), "tri_ntxent_bwd")
    return dza, dzb


TIMELINE = None      # tools/step_timeline.py sets {"buf": int64 device tensor, "names": []}: stamp() then timestamps the stream


def stamp(name: str):
    """Device-side timestamp of this point of the current stream (no-op unless a timeline is being recorded)."""
    tl = TIMELINE
    if tl is None:
        return
    idx = len(tl["names"])
    if idx >= tl["buf"].numel():
        return
    tl["names"].append(name)
    check(lib().tri_debug_stamp(tl["buf"].data_ptr() + 8 * idx, stream()), "tri_debug_stamp")


DEBUG_KEEP = None            # dict: debugging tools park intermediate tensors here (tools/r6/determinism.py)
_ONES = {}


def one(device) -> torch.Tensor:
    """A cached 0-d fp32 one on `device`: `loss.backward(gradient=ops.one(loss.device))` spares autograd's fill launch per step."""
    device = torch.device(device)
    t = _ONES.get(device)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            # created inside a capture its fill kernel would only be RECORDED: an eager use before the first replay would read
            # uninitialised memory.  FusedAdam.prepare() creates it ahead of any capture; anything else must call ops.one() eagerly first.
            raise RuntimeError("ops.one(): first use inside a HIP-graph capture - call ops.one(device) (or FusedAdam.prepare()) before capturing")
        t = torch.ones((), dtype=torch.float32, device=device)
        _ONES[device] = t
    return t


def ntxent_multi_supported(B: int, D: int, M: int) -> bool:
    return M in (2, 3) and 1 <= B <= 512 and 4 <= D <= 2048 and D % 4 == 0


def ntxent_multi_fwd(zs, temperature, alpha, norm=True):
    """Every pair of the M = 2 / 3 embeddings `zs` at once: (losses [P + 1] = pair losses in combination order, then their
    sum; workspace for ntxent_multi_bwd)."""
    M = len(zs)
    B, D = zs[0].shape
    P = M * (M - 1) // 2
    losses = torch.empty((P + 1,), dtype=torch.float32, device=zs[0].device)
    nbytes = lib().tri_ntxent_multi_workspace(M, B, D)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=zs[0].device)
    zp = (_C.C.c_void_p * M)(*[ptr(_f32(z)) for z in zs])
    check(lib().tri_ntxent_multi_fwd(zp, M, B, D, float(temperature), float(alpha), 1 if norm else 0, ptr(losses), ptr(ws), nbytes,
                                     stream()), "tri_ntxent_multi_fwd")
    return losses, ws


def ntxent_multi_bwd(zs, ws, temperature, alpha, norm=True, dpairs=None, dtotal=None):
    """Gradients of all M embeddings in one launch; dpairs (list of 0-d device tensors or None) / dtotal are the upstream
    gradients of the pair losses / of their sum."""
    M = len(zs)
    B, D = zs[0].shape
    P = M * (M - 1) // 2
    dzs = [torch.empty_like(z) for z in zs]
    zp = (_C.C.c_void_p * M)(*[ptr(_f32(z)) for z in zs])
    dp = (_C.C.c_void_p * P)(*[ptr(d) for d in dpairs]) if dpairs is not None else None
    dzp = (_C.C.c_void_p * M)(*[ptr(d) for d in dzs])
    check(lib().tri_ntxent_multi_bwd(zp, M, B, D, float(temperature), float(alpha), 1 if norm else 0, dp, ptr(dtotal), dzp, ptr(ws),
                                     ws.numel(), stream()), "tri_ntxent_multi_bwd")
    return dzs


def ntxent_fwd_bwd(za, zb, temperature, alpha, norm=True, want_grad=True):
    B, D = za.shape
    loss = torch.empty((), dtype=torch.float32, device=za.device)
    dza = torch.empty_like(za) if want_grad else None
    dzb = torch.empty_like(zb) if want_grad else None
    nbytes = lib().tri_ntxent_workspace(B, D)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=za.device)
    check(lib().tri_ntxent_fwd_bwd(ptr(_f32(za.contiguous())), ptr(_f32(zb.contiguous())), B, D, float(temperature), float(alpha),
                                   1 if norm else 0, ptr(loss), ptr(dza), ptr(dzb), ptr(ws), nbytes, stream()), "tri_ntxent_fwd_bwd")
    return loss, dza, dzb


# ------------------------------------------------------------------------------------------------ Adam
def adam_tick(step):
    check(lib().tri_adam_tick(ptr(step), stream()), "tri_adam_tick")


def adam_guard(g, step):
    """Per-step overflow guard over a flat gradient: flags the attempt in step[2] when g holds an inf / NaN (call before adam_tick)."""
    check(lib().tri_adam_guard(ptr(_f32(g)), g.numel(), ptr(step), stream()), "tri_adam_guard")


def adam_guard_segments(grad_ptrs, grad_starts, n, step):
    check(lib().tri_adam_guard_segments(ptr(grad_ptrs), ptr(grad_starts), grad_starts.numel(), n, ptr(step), stream()), "tri_adam_guard_segments")


def adam_step_segments(p, grad_ptrs, grad_starts, m, v, step, lr, b1, b2, eps, wd, gscale=1.0, lr_dev=None):
    """Fused Adam over the flat buffers with the gradients read in place through a device (pointer, start) table.
    lr_dev (1-element fp32 device tensor) overrides lr at run time: a captured graph then follows an LR schedule."""
    check(lib().tri_adam_step_segments(ptr(p), ptr(grad_ptrs), ptr(grad_starts), grad_starts.numel(), ptr(m), ptr(v), p.numel(), ptr(step),
                                       lr, ptr(lr_dev), b1, b2, eps, wd, gscale, stream()), "tri_adam_step_segments")


def adam_step(p, g, m, v, step, lr, b1, b2, eps, wd, gscale=1.0, lr_dev=None):
    check(lib().tri_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(step), lr, ptr(lr_dev), b1, b2, eps, wd, gscale, stream()),
          "tri_adam_step")
