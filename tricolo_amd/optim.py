"""Fused Adam on the gfx950 kernel: torch.optim.Adam semantics (L2 weight decay folded into the gradient, bias
correction, eps added after the sqrt) as instantiated by /root/reference/config/config.yaml:50-53 through
tricolo_net.py:43-44.  Select it with ``optimizer._target_: tricolo_amd.optim.FusedAdam`` (the default of
tricolo_amd/config/config.yaml); ``torch.optim.Adam`` keeps working on the same parameters.

MI355X design: all parameters are re-bound (once) to views of ONE flat fp32 buffer; each step concatenates the
gradients into one flat buffer (a single copy kernel), optionally hands that buffer to the data-parallel all-reduce
(one collective for the whole model: xGMI rings are per-link bound, so fewer and bigger), and updates everything with
ONE kernel launch.  The step counter AND the learning rate live on the device, so the optimizer step is HIP-graph
capturable and a replayed graph follows an LR schedule (the reference's LrDecayCallback, tricolo/callback/lr_decay.py):
change ``param_groups[0]["lr"]`` as usual; eager steps pick it up by themselves, graph replays after ``sync_lr()``.

``state_dict()`` / ``load_state_dict()`` speak torch.optim.Adam's format (per-parameter ``step``, ``exp_avg``,
``exp_avg_sq``), so Lightning's ``optimizer_states`` checkpoints resume in either optimizer.
"""
import os

import torch

from . import ops


_GUARD_NOTE = os.environ.get("TRICOLO_GUARD_NOTE", "1") != "0"     # A/B switch (round 6): 0 = the guard scans every gradient itself


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, flatten=True, guard=True):
        """guard: scan every step's gradient for inf / NaN first (one extra read of the gradients, ~10 us for 18.5 M parameters) and skip
        the WHOLE step when one is found - parameters, both moments and the step counter stay, `skipped_steps()` counts it (what
        torch.cuda.amp.GradScaler does with an overflowed step; the f16 mode's activation gradients can overflow).  guard=False keeps
        only the per-element fallback of the update kernels (a non-finite element leaves its own p, m, v alone).  Either way this
        DIFFERS from torch.optim.Adam, which would write the NaN into the weights (the reference's behaviour: a NaN surfaces as a NaN loss)."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._guard = bool(guard)
        self._flatten = flatten and len(self.param_groups) == 1
        self._flat_p = self._flat_m = self._flat_v = self._step_dev = self._lr_dev = None
        self._lr_host = None
        self._params = None
        self._pending_state = None
        self._captures = 0
        self._zg_event = None
        self._seg_side = None

    def zero_grad(self, set_to_none: bool = True):
        """torch.optim.Optimizer.zero_grad; inside a HIP-graph capture it also marks the point the step's gradient-pointer table may be
        uploaded from (see _update_from_segments)."""
        super().zero_grad(set_to_none=set_to_none)
        self._zg_event = None
        # lend the gradient writers this optimizer's overflow record (ops.set_guard_note): what they vouch for is left out of the scan below
        ops.set_guard_note(self._step_dev if (_GUARD_NOTE and self._guard and self._step_dev is not None and getattr(self, "_seg_ok", False)) else None)
        if self._step_dev is not None and torch.cuda.is_current_stream_capturing():
            self._zg_event = torch.cuda.Event()
            self._zg_event.record()

    # ------------------------------------------------------------------ state
    def prepare(self, captures: int = 0):
        """Allocate state (and flatten the parameters) ahead of the first step / HIP-graph capture.  captures: number of HIP-graph
        captures of the step the caller is going to make (each keeps one pinned gradient-pointer table; 32 are provided anyway)."""
        self._captures = max(int(captures), getattr(self, "_captures", 0))
        if self._step_dev is not None:
            while hasattr(self, "_seg_capture_pool") and len(self._seg_capture_pool) + len(self._seg_captured) < self._captures:
                self._seg_capture_pool.append(torch.zeros((3 * len(self._params),), dtype=torch.int64).pin_memory())
            return
        params = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if not params:
            return
        for p in params:
            if not p.is_cuda:
                raise RuntimeError("FusedAdam: parameter is not on a GPU (no CPU fallback)")
            if p.dtype != torch.float32:
                raise RuntimeError("FusedAdam: fp32 master parameters expected")
        dev = params[0].device
        self._params = params
        # [0] applied steps, [1] skipped non-finite elements (fallback), [2] attempt flagged by the guard, [3] steps skipped whole (misc.hip)
        self._step_dev = torch.zeros((4,), dtype=torch.int32, device=dev)
        ops.one(dev)                                                             # the cached loss-gradient scalar: never first created inside a capture
        self._lr_dev = torch.zeros((1,), dtype=torch.float32, device=dev)
        self.sync_lr()
        if self._flatten:
            sizes = [p.numel() for p in params]
            flat = torch.empty((sum(sizes),), dtype=torch.float32, device=dev)
            off = 0
            with torch.no_grad():
                for p, n in zip(params, sizes):
                    flat[off:off + n].copy_(p.detach().reshape(-1))
                    p.data = flat[off:off + n].view(p.shape)         # same Parameter object, storage now inside `flat`
                    off += n
            self._flat_p = flat
            self._flat_m = torch.zeros_like(flat)
            self._flat_v = torch.zeros_like(flat)
            # segment table for the in-place gradient read (single-GPU step): flat start of every parameter, and a pinned
            # host + device pair for the gradient pointers, refreshed per step (captured once under a HIP graph)
            self._seg_ok = all(n % 4 == 0 for n in sizes) and len(sizes) <= 1024
            if self._seg_ok:
                starts = [0]
                for n in sizes[:-1]:
                    starts.append(starts[-1] + n)
                self._seg_start = torch.tensor(starts, dtype=torch.int64, device=dev)
                # eager steps: a small ring of pinned staging buffers, each guarded by an event (the stream may lag the host by
                # many kernels, so a buffer is only refilled after its previous copy ran); a HIP-graph capture gets a staging
                # buffer of its own that is never written again (the captured copy re-reads it on every replay)
                # (table layout, 3 n entries: gradient pointers | pointers of the segments the guard scans | their starts in the scan's order)
                self._seg_sizes = sizes
                self._seg_ring = [[torch.zeros((3 * len(sizes),), dtype=torch.int64).pin_memory(), None] for _ in range(4)]
                self._seg_ring_i = 0
                # (bench.py captures one graph per resident batch - 8 by default -, tests and re-captures add more: 32 tables of a few KB;
                #  prepare(captures=n) sizes it for a caller that knows better; an exhausted pool is reported, never silent)
                self._seg_capture_pool = [torch.zeros((3 * len(sizes),), dtype=torch.int64).pin_memory() for _ in range(max(32, self._captures))]
                self._seg_captured = []
                self._seg_tab = torch.zeros((3 * len(sizes),), dtype=torch.int64, device=dev)
                self._seg_ptr = self._seg_tab[:len(sizes)]
                self._seg_keep = None
            off = 0
            for p, n in zip(params, sizes):                          # torch-compatible per-parameter state views
                self.state[p]["exp_avg"] = self._flat_m[off:off + n].view(p.shape)
                self.state[p]["exp_avg_sq"] = self._flat_v[off:off + n].view(p.shape)
                off += n
        else:
            for p in params:
                self.state[p]["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                self.state[p]["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        if self._pending_state is not None:                          # load_state_dict() came before the first step
            sd, self._pending_state = self._pending_state, None
            self._restore(sd)

    def nonfinite_skipped(self) -> int:
        """Number of gradient elements the kernels refused so far because they were inf / NaN (f16 activation-gradient overflow guard:
        such an element leaves its parameter and both moments untouched - csrc/misc.hip adam_note_bad).  Reads a device counter
        (synchronises); 0 in a healthy run.  A training loop should look at it now and then and lower ops.F16_GRAD_SCALE / switch to
        the bf16x3 mode when it grows."""
        return 0 if self._step_dev is None else int(self._step_dev[1].item())

    def skipped_steps(self) -> int:
        """Optimizer steps skipped WHOLE because the guard pass found an inf / NaN in that step's gradient (reads a device counter:
        synchronises - look at it at a cheap cadence, e.g. once per epoch, and lower ops.F16_GRAD_SCALE / switch to bf16x3 if it grows)."""
        return 0 if self._step_dev is None else int(self._step_dev[3].item())

    def sync_lr(self):
        """Copy param_groups[0]['lr'] to the device scalar the kernels read.  Eager steps call it themselves; call it after
        changing the learning rate when the step is replayed from a HIP graph (it cannot run inside a capture)."""
        if self._lr_dev is None:
            return
        lr = float(self.param_groups[0]["lr"])
        if lr == self._lr_host:
            return
        if torch.cuda.is_current_stream_capturing():
            if self._lr_host is None:
                raise RuntimeError("FusedAdam: call prepare() before capturing the step into a HIP graph")
            return                                                   # picked up by the next sync_lr() outside the capture
        self._lr_dev.fill_(lr)
        self._lr_host = lr

    # ------------------------------------------------------------------ checkpoint format of torch.optim.Adam
    def state_dict(self):
        """torch.optim.Adam's layout: state[i] = {'step': 0-d fp32 tensor, 'exp_avg', 'exp_avg_sq'} (what Lightning stores
        under ``optimizer_states`` and ``trainer.fit(ckpt_path=...)`` restores, train.py:41-45)."""
        sd = super().state_dict()
        if self._step_dev is not None:
            step = self._step_dev[0].to(torch.float32).reshape(())
            # the packed per-parameter dicts ARE self.state's own and the moments are views of the flat buffers the kernels
            # update: hand out copies (an optimizer that load_state_dict()s this dict in the same process keeps the tensors)
            sd["state"] = {k: {"step": step.clone(), "exp_avg": v["exp_avg"].clone(), "exp_avg_sq": v["exp_avg_sq"].clone()}
                           for k, v in sd["state"].items()}
        return sd

    def _restore(self, sd):
        params = [p for g in self.param_groups for p in g["params"]]
        ids = [i for g in sd["param_groups"] for i in g["params"]]
        steps = set()
        with torch.no_grad():
            for p, i in zip(params, ids):
                st = sd["state"].get(i)
                if st is None or p not in self.state:
                    continue
                self.state[p]["exp_avg"].copy_(st["exp_avg"])        # in place: the kernels read the flat buffers behind these views
                self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
                if "step" in st:
                    steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise RuntimeError(f"FusedAdam keeps ONE step counter; the checkpoint holds different per-parameter steps {sorted(steps)}")
            if steps:
                self._step_dev[0].fill_(steps.pop())
            # the guard's record speaks in ATTEMPT numbers (applied + skipped): [2] = the attempt it flagged last (atomicMax), [3] = steps
            # skipped whole.  Rewinding [0] alone would leave a stale flag above every attempt of the restored run until it catches up -
            # those steps would fall back to the per-element guard, and the attempt that reaches the old flag would be skipped although
            # its gradient is finite.  A restored optimizer starts a fresh record.
            self._step_dev[1:].zero_()
        for g, gs in zip(self.param_groups, sd["param_groups"]):
            for k, v in gs.items():
                if k != "params":
                    g[k] = v
        self._lr_host = None
        self.sync_lr()

    def load_state_dict(self, state_dict):
        if len(state_dict["param_groups"]) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        n_here = [len(g["params"]) for g in self.param_groups]
        if [len(g["params"]) for g in state_dict["param_groups"]] != n_here:
            raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
        params = [p for g in self.param_groups for p in g["params"]]
        if params and all(p.is_cuda for p in params):
            self.prepare()
        if self._step_dev is None:                                   # parameters not on the GPU yet: applied by prepare()
            self._pending_state = state_dict
            return
        self._restore(state_dict)

    def flat_grad(self):
        """All gradients as one contiguous fp32 buffer in parameter order (zeros for parameters without a gradient)."""
        parts = []
        for p in self._params:
            g = p.grad
            parts.append(g.reshape(-1) if g is not None else torch.zeros(p.numel(), dtype=torch.float32, device=p.device))
        return torch.cat(parts)

    def _update_from_segments(self, group, b1, b2, grad_scale) -> bool:
        """Single-GPU step without the concatenation pass: the kernel reads every gradient where autograd left it."""
        grads, ptrs = [], []
        for p in self._params:
            g = p.grad
            if g is not None:
                if g.dtype != torch.float32 or not g.is_cuda:
                    return False
                if not g.is_contiguous():
                    g = g.contiguous()
                if g.data_ptr() % 16:
                    return False
            grads.append(g)
            ptrs.append(g.data_ptr() if g is not None else 0)
        n = len(ptrs)
        # the guard's scan list: every gradient no writer vouched for (ops.guard_noted: the grouped weight-gradient reduce notes the
        # inf / NaN it stores itself), compacted - the scan's grid and its binary search then cover those elements only
        gptrs, gstarts, gtotal = [], [], 0
        for a, size, g in zip(ptrs, self._seg_sizes, grads):
            if a and not ops.guard_noted(g, self._step_dev):
                gptrs.append(a)
                gstarts.append(gtotal)
                gtotal += size
        table = ptrs + gptrs + [0] * (n - len(gptrs)) + gstarts + [0] * (n - len(gstarts))
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            if not self._seg_capture_pool:                           # (pinned memory cannot be allocated inside a capture)
                if not getattr(self, "_pool_warned", False):
                    import warnings
                    warnings.warn("FusedAdam: no pinned gradient-pointer table left for this HIP-graph capture (more than "
                                  f"{len(self._seg_captured)} captures); the capture records the packed-gradient path instead - "
                                  "call prepare(captures=n) before capturing")
                    self._pool_warned = True
                return False
            host = self._seg_capture_pool.pop()
            host.copy_(torch.tensor(table, dtype=torch.int64))
            self._seg_captured.append(host)                          # owned by the graph from now on
        else:
            slot = self._seg_ring[self._seg_ring_i]
            self._seg_ring_i = (self._seg_ring_i + 1) % len(self._seg_ring)
            if slot[1] is not None:
                slot[1].synchronize()
            host = slot[0]
            host.copy_(torch.tensor(table, dtype=torch.int64))
        self._seg_keep = grads                                       # keep temporaries alive until the next step
        ev, self._zg_event = getattr(self, "_zg_event", None), None
        if capturing and ev is not None:
            # (round 6) captured step: the pointer table's copy node depends on nothing the step computes - it is issued here but hangs on
            # the event zero_grad() recorded (same capture), so it runs long before the backward ends instead of sitting, with its
            # dependent-node gap, between the last gradient kernel and the guard on the serial tail of the step
            if self._seg_side is None:
                self._seg_side = torch.cuda.Stream()
            main = torch.cuda.current_stream()
            self._seg_side.wait_event(ev)
            with torch.cuda.stream(self._seg_side):
                self._seg_tab.copy_(host, non_blocking=True)
            main.wait_stream(self._seg_side)
        else:
            self._seg_tab.copy_(host, non_blocking=True)
        if not capturing:
            slot[1] = torch.cuda.Event()
            slot[1].record()
        if self._guard and gptrs:
            ops.adam_guard_segments(self._seg_tab[n:n + len(gptrs)], self._seg_tab[2 * n:2 * n + len(gptrs)], gtotal, self._step_dev)
        # (round 6: the tick folded into the guard launch's last workgroup - one node less - was built and measured: 4,096 workgroups taking a
        #  ticket on one address cost 110 us; the kernel-argument form of the pointer table was dropped with it)
        ops.adam_tick(self._step_dev)                       # step += 1 on the device (or skipped += 1), once per optimizer step
        ops.adam_step_segments(self._flat_p, self._seg_ptr, self._seg_start, self._flat_m, self._flat_v, self._step_dev, group["lr"],
                               b1, b2, group["eps"], group["weight_decay"], grad_scale, lr_dev=self._lr_dev)
        return True

    @torch.no_grad()
    def apply_flat(self, g: torch.Tensor, grad_scale: float = 1.0):
        """The update half of step() for an already packed (and, data-parallel, already all-reduced) flat gradient.
        parallel.GraphedDPStep captures flat_grad() and apply_flat() into separate HIP graphs around the eager all-reduce."""
        if not self._flatten:
            raise RuntimeError("apply_flat needs flatten=True")
        group = self.param_groups[0]
        b1, b2 = group["betas"]
        self.sync_lr()
        if self._guard:
            ops.adam_guard(g, self._step_dev)
        ops.adam_tick(self._step_dev)
        ops.adam_step(self._flat_p, g, self._flat_m, self._flat_v, self._step_dev, group["lr"], b1, b2, group["eps"],
                      group["weight_decay"], grad_scale, lr_dev=self._lr_dev)

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, reduce_fn=None):
        """reduce_fn(flat_grad) (optional) runs between gradient packing and the update - the data-parallel hook."""
        loss = closure() if closure is not None else None
        self.prepare()
        if self._step_dev is None:
            return loss
        group = self.param_groups[0]
        b1, b2 = group["betas"]
        self.sync_lr()
        ops.stamp("adam.start")
        if self._flatten:
            # (the segment reader returns False BEFORE it launches anything when it cannot take the step)
            if reduce_fn is None and self._seg_ok and self._update_from_segments(group, b1, b2, grad_scale):
                ops.stamp("adam.end")
                return loss
            if any(p.grad is None for p in self._params):
                # torch.optim.Adam skips such parameters entirely (no weight decay, no moment decay); the flat kernel cannot
                raise RuntimeError("FusedAdam(flatten=True): a parameter has no gradient; the packed-gradient path (data-parallel "
                                   "reduce_fn, or parameter sizes not multiples of 4) needs one for every parameter")
            g = self.flat_grad()
            if reduce_fn is not None:
                reduce_fn(g)
            if self._guard:
                ops.adam_guard(g, self._step_dev)               # (after the all-reduce: every rank takes the same decision)
            ops.adam_tick(self._step_dev)
            ops.adam_step(self._flat_p, g, self._flat_m, self._flat_v, self._step_dev, group["lr"], b1, b2, group["eps"],
                          group["weight_decay"], grad_scale, lr_dev=self._lr_dev)
            return loss
        if reduce_fn is not None:
            raise RuntimeError("reduce_fn needs flatten=True")
        if self._guard:                                          # per-parameter path (never graph-captured): guard every gradient first
            for group in self.param_groups:
                for p in group["params"]:
                    if p.grad is not None:
                        ops.adam_guard(p.grad if p.grad.is_contiguous() else p.grad.contiguous(), self._step_dev)
        ops.adam_tick(self._step_dev)
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adam_step(p, g, st["exp_avg"], st["exp_avg_sq"], self._step_dev, group["lr"], b1, b2, group["eps"],
                              group["weight_decay"], grad_scale)             # per-group lr by value: this path is never graph-captured
        return loss
