"""Data-parallel pieces of the TriCoLo step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on MI355X, "gloo" in the CPU tests).

The reference has no distributed code (Lightning's default DDP: local-batch negatives, un-synchronised BatchNorm,
SURVEY.md section 2.3).  The north star adds ONE exchange step: an all-gather of the modality embeddings so every
rank computes NT-Xent over the full global batch.  Design:
  * forward: the 2-3 [B_local, 512] matrices are packed into one [B_local, 512*n_mod] buffer -> a single
    all_gather_into_tensor (192 KB/rank at B_local=32: latency-bound, so one message instead of three);
  * every rank evaluates the identical global loss; backward hands each rank the rows of d(loss)/d(z) that belong to
    its own samples - no reduce-scatter is needed because all ranks hold the same dS;
  * parameter gradients are SUM-all-reduced in one flat bucket per call (d L_global / d theta = sum over ranks of the
    local-path gradients); BatchNorm keeps per-rank statistics like the reference's DDP default.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import ops


def is_dist() -> bool:
    """True when the data-parallel exchange steps must run.  TRICOLO_FORCE_DIST=1 also takes that path in a world of one
    (how the graph-split step below is exercised on a single-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("TRICOLO_FORCE_DIST", "0") == "1"


def _host_staged(t: torch.Tensor) -> bool:
    """Device tensors under the gloo backend (several ranks sharing ONE GPU in the tests, or a box without RCCL) cross the
    process boundary through host memory; under "nccl" (= RCCL over xGMI) the collective takes the device buffer itself."""
    return t.is_cuda and dist.get_backend() == "gloo"


class _Done:
    """Handle of a collective that already completed (the host-staged path is synchronous)."""

    def wait(self):
        return True


def all_gather_rows(out: torch.Tensor, x: torch.Tensor) -> None:
    """out[world * rows, ...] = the ranks' x[rows, ...] in rank order (all_gather_into_tensor)."""
    if _host_staged(x):
        ho = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(ho, x.detach().cpu().contiguous())
        out.copy_(ho)
        return
    dist.all_gather_into_tensor(out, x.contiguous())


def all_reduce_sum(t: torch.Tensor, async_op: bool = False):
    """In-place SUM all-reduce of a (flat, contiguous) tensor; returns a handle with wait() when async_op."""
    if _host_staged(t):
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
        return _Done() if async_op else None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


class _AllGatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        world = dist.get_world_size()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        all_gather_rows(out, x)
        ctx.rows = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        r = dist.get_rank()
        return dout[r * ctx.rows:(r + 1) * ctx.rows].contiguous()


def gather_embeddings(output_dict: dict) -> dict:
    """{name: [B_local, D]} -> {name: [B_global, D]} with one fused all-gather (rank-major row order)."""
    if not is_dist():
        return output_dict
    keys = list(output_dict.keys())
    packed = torch.cat([output_dict[k] for k in keys], dim=1)
    full = _AllGatherRows.apply(packed)
    out, off = {}, 0
    for k in keys:
        d = output_dict[k].shape[1]
        out[k] = full[:, off:off + d].contiguous()
        off += d
    return out


def allreduce_gradients(params, op=None) -> None:
    """SUM-all-reduce every .grad in one flat bucket (bigger, fewer collectives: xGMI rings are per-link bound)."""
    if not is_dist():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    if op is None or op == dist.ReduceOp.SUM:
        all_reduce_sum(flat)
    else:
        dist.all_reduce(flat, op=op)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def allreduce_flat(flat: torch.Tensor) -> None:
    """In-place SUM all-reduce of one flat gradient bucket (hook for FusedAdam.step(reduce_fn=...))."""
    if is_dist():
        all_reduce_sum(flat)


class BackwardSplit:
    """Where the backward pass can be cut so that the gradient all-reduce of everything ABOVE the cut overlaps the backward of
    what lies BELOW it.  The forward pass routes the activation at the cut through gate(), which hands the upper part a detached
    leaf: stage 1 (an ordinary backward from the loss) then ends there with d loss / d cut in the leaf's .grad and never touches
    the graph below; stage 2 is an ordinary backward from the cut tensor with that gradient.  A plain loss.backward() on a
    module with an installed split would silently skip the lower part - the leaf's hook raises instead.  `late_params` belong
    to the part below (they get their gradients last).  Several gates may be open in one forward pass (stage 2 then runs
    from all cuts at once).  For TriCoLoNet on the HIP modules there are two: inside the image tower between layer2 and layer3
    (MVCNNEncoder installs it when it is given a split) and at the voxel tower's output (TriCoLoNet.forward).  Stage 1 = text
    tower + image layer3-4 + heads (68 % of the gradient bytes of Tri(I+V) at 32^3: 12.6 of 18.5 M parameters, reduced while stage 2
    runs); stage 2 = image stem + layer1-2 (0.68 M) beside the whole voxel tower (5.2 M) on its own stream - measured on one GPU, a single gate inside the image tower cost 0.43 ms per
    step because the voxel tower's backward, no longer hidden under the image tower's, then bounded stage 1."""

    def __init__(self, net, late_params):
        self.late_params = list(late_params)
        late = {id(p) for p in self.late_params}
        self.early_params = [p for p in net.parameters() if p.requires_grad and id(p) not in late]
        self.defer, self.cuts, self.leaves, self._open = False, [], [], False

    def _check(self, g):
        if not self.defer:
            raise RuntimeError("this module has a parallel.BackwardSplit installed: run its backward through "
                               "parallel.backward_overlapped / dp_training_step(split=...) (stage 1 + stage 2), not loss.backward()")
        return g

    def gate(self, x):
        if not self._open:                                   # first gate of a new forward pass
            self.cuts, self.leaves, self._open = [], [], True
        leaf = x.detach().requires_grad_()
        leaf.register_hook(self._check)
        self.cuts.append(x)
        self.leaves.append(leaf)
        return leaf

    @staticmethod
    def for_net(net):
        """The split of a tricolo_amd TriCoLoNet with an MVCNNEncoder image tower (None when there is nothing to split)."""
        enc = getattr(net, "image_encoder", None)
        if enc is None or not hasattr(enc, "_lower_params"):
            return None
        late = list(enc._lower_params())
        vox = getattr(net, "voxel_encoder", None)
        if vox is not None:
            late += [p for p in vox.parameters() if p.requires_grad]
        split = BackwardSplit(net, late)
        enc.__dict__["split"] = split
        net.__dict__["dp_split"] = split                     # TriCoLoNet.forward gates the voxel tower's output
        return split


def _runs(params, order):
    """Contiguous runs [(start, end, [params])] of `params` inside the flat layout given by `order` (list of parameters)."""
    want = {id(p) for p in params}
    runs, off, cur = [], 0, None
    for p in order:
        n = p.numel()
        if id(p) in want:
            if cur is None:
                cur = [off, off, []]
            cur[1] = off + n
            cur[2].append(p)
        elif cur is not None:
            runs.append(tuple(cur))
            cur = None
        off += n
    if cur is not None:
        runs.append(tuple(cur))
    return runs, off


def _pack(params, out):
    # every parameter of a bucket needs a gradient: the flat Adam kernel updates ALL of them (weight decay, moment decay), which is not
    # what torch.optim.Adam does for a parameter without one - the same rule FusedAdam.step enforces on the single-bucket path
    if any(p.grad is None for p in params):
        raise RuntimeError("parallel: a parameter of the gradient bucket has no gradient (the packed-gradient path needs one for every "
                           "parameter; freeze it with requires_grad_(False) instead)")
    torch.cat([p.grad.reshape(-1) for p in params], out=out)


def backward_stage1(root, grad_root, split: BackwardSplit):
    """Backward from `root` down to the gate: everything above the cut gets its .grad, the gate keeps d root / d cut."""
    if not split.leaves:
        raise RuntimeError("BackwardSplit: the forward pass did not go through a gate (is the split installed in the module?)")
    split._open = False
    split.defer = True
    try:
        torch.autograd.backward(root, grad_root)
    finally:
        split.defer = False


def backward_stage2(split: BackwardSplit):
    """The part below the gate, from the gradient stage 1 left there."""
    torch.autograd.backward(split.cuts, [l.grad for l in split.leaves])
    split.cuts, split.leaves = [], []


def backward_overlapped(total, split: BackwardSplit, order, flat=None):
    """Backward of `total` in two stages around the split's gate, the SUM all-reduce of the early parameters' gradients
    issued asynchronously between them.  `order` = the parameters in flat-buffer order (FusedAdam's); returns the flat,
    fully reduced gradient (every rank holds d L_global / d theta).  Both stages are ordinary autograd.backward() calls."""
    early_runs, total_n = _runs(split.early_params, order)
    late_runs, _ = _runs(split.late_params, order)
    if flat is None:
        flat = torch.empty((total_n,), dtype=torch.float32, device=order[0].device)
    handles = []
    backward_stage1(total, None, split)
    for s, e, ps in early_runs:
        _pack(ps, flat[s:e])
        if is_dist():
            handles.append(all_reduce_sum(flat[s:e], async_op=True))                            # runs under stage 2
    backward_stage2(split)
    for s, e, ps in late_runs:
        _pack(ps, flat[s:e])
        if is_dist():
            handles.append(all_reduce_sum(flat[s:e], async_op=True))
    for h in handles:
        h.wait()
    return flat


def dp_training_step(net, batch, optimizer=None, split: BackwardSplit | None = None):
    """One data-parallel training step of TriCoLoNet: local towers -> gathered embeddings -> global losses ->
    backward -> gradient all-reduce (-> optimizer step).  Returns the loss dict (identical on every rank).
    With `split` (BackwardSplit.for_net) and a flat optimizer the all-reduce of the early bucket overlaps the rest of the backward."""
    out = net(batch)
    out = gather_embeddings(out)
    losses = net._calculate_losses(out, "train_loss")
    if optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    if split is not None and optimizer is not None and getattr(optimizer, "_flatten", False):
        # the two-stage backward whenever a split is installed (its gate refuses a plain backward); in a world of one the buckets
        # are simply not reduced
        optimizer.prepare()
        flat = backward_overlapped(losses["train_loss/total_loss"], split, optimizer._params)
        optimizer.apply_flat(flat)
        return losses
    losses["train_loss/total_loss"].backward(gradient=ops.one(losses["train_loss/total_loss"].device))
    if hasattr(net, "join_side_streams"):
        net.join_side_streams()                                              # (every tower's gradient kernels are ordered before the optimizer's)
    if optimizer is not None and getattr(optimizer, "_flatten", False):
        optimizer.step(reduce_fn=allreduce_flat if is_dist() else None)    # one bucket: pack -> all-reduce -> fused update
        return losses
    allreduce_gradients(list(net.parameters()))
    if optimizer is not None:
        optimizer.step()
    return losses


class GraphedDPStep:
    """The data-parallel step as HIP graphs with the collectives issued eagerly between them:

        graph A   towers forward -> packed local embeddings [B_local, 512 * n_mod]
        eager     all_gather_into_tensor(full, packed)                      (RCCL over xGMI)
        graph B   global NT-Xent on `full`, backward to this rank's rows, towers backward, flat gradient pack
        eager     all_reduce(flat gradient, SUM)
        graph C   fused Adam on the flat buffers

    With `split` (BackwardSplit.for_net) graph B is cut in two at the image tower's layer2 / layer3 boundary:

        graph B1  loss, backward of the text / voxel towers and of the image tower's upper half, pack of those gradients
        eager     all_reduce(early ranges of the flat gradient, async)      <- 68 % of the bytes (Tri 32^3), runs UNDER graph B2
        graph B2  backward of stem + layer1 + layer2, pack of the late range
        eager     all_reduce(late range, async); wait for both

    Collectives are deliberately NOT captured (RCCL under hipGraph capture cannot be validated on a one-GPU box); the
    ~400 kernel launches of the step are, so a replayed step costs a few graph launches + the collectives on the host
    instead of ~7 ms of Python launch overhead.  The graphs share one memory pool: activations saved by graph A's forward
    are consumed by graph B's backward.  One instance per resident batch (inputs are static)."""

    def __init__(self, net, optimizer, batch, split: BackwardSplit | None = None):
        if not getattr(optimizer, "_flatten", False):
            raise RuntimeError("GraphedDPStep needs FusedAdam(flatten=True)")
        self.net, self.opt, self.split = net, optimizer, split
        world, rank = dist.get_world_size(), dist.get_rank()
        pool = torch.cuda.graph_pool_handle()
        self.gA, self.gB, self.gC = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self.gB2 = torch.cuda.CUDAGraph() if split is not None else None
        optimizer.zero_grad(set_to_none=True)
        optimizer.prepare()
        # thread_local: the process group's watchdog thread keeps polling events of earlier collectives while this thread
        # captures; under the default global mode that poll is an illegal call and aborts the process
        mode = dict(pool=pool, capture_error_mode="thread_local")
        with torch.cuda.graph(self.gA, **mode):
            out = net(batch)
            keys = list(out.keys())
            dims = [out[k].shape[1] for k in keys]
            packed = torch.cat([out[k] for k in keys], dim=1)
        rows = packed.shape[0]
        self.packed = packed
        self.full = torch.zeros((world * rows, packed.shape[1]), dtype=packed.dtype, device=packed.device)
        order = optimizer._params
        if split is not None:
            self.early_runs, total_n = _runs(split.early_params, order)
            self.late_runs, _ = _runs(split.late_params, order)
            self.flat = torch.zeros((total_n,), dtype=torch.float32, device=packed.device)
        with torch.cuda.graph(self.gB, **mode):
            leaf = self.full.detach().requires_grad_()
            glob, off = {}, 0
            for k, d in zip(keys, dims):
                glob[k] = leaf[:, off:off + d].contiguous()
                off += d
            losses = net._calculate_losses(glob, "train_loss")
            total = losses["train_loss/total_loss"]
            (dfull,) = torch.autograd.grad(total, leaf)                 # identical on every rank: no reduce-scatter needed
            dlocal = dfull[rank * rows:(rank + 1) * rows].contiguous()
            if split is None:
                packed.backward(dlocal)
                self.flat = optimizer.flat_grad()
            else:
                backward_stage1(packed, dlocal, split)
                for s_, e_, ps in self.early_runs:
                    _pack(ps, self.flat[s_:e_])
            self.loss = total.detach()
        if split is not None:
            with torch.cuda.graph(self.gB2, **mode):
                backward_stage2(split)
                for s_, e_, ps in self.late_runs:
                    _pack(ps, self.flat[s_:e_])
        with torch.cuda.graph(self.gC, **mode):
            optimizer.apply_flat(self.flat)

    def replay_timed(self):
        """One replay with HIP events between its phases on the current stream -> (loss, events); GraphedDPStep.phase_ms(events) after a
        synchronize gives {phase: ms}.  A collective issued through torch.distributed runs on the process group's own stream; the
        current stream waits for it when the (synchronous) call returns / at handle.wait(), so the event recorded behind it fires
        when the collective has completed: `all_gather` and `all_reduce` are the EXPOSED times of the two exchange steps (bench.py
        prints their medians per rank into the JSON line, so the first multi-GPU run yields the breakdown)."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record()
        self.gA.replay()
        ev[1].record()
        all_gather_rows(self.full, self.packed)
        ev[2].record()
        self.gB.replay()
        if self.split is None:
            ev[3].record()
            all_reduce_sum(self.flat)
        else:
            hs = [all_reduce_sum(self.flat[s:e], async_op=True) for s, e, _ in self.early_runs]
            self.gB2.replay()
            ev[3].record()                                              # (the early ranges are being reduced under graph B2)
            hs += [all_reduce_sum(self.flat[s:e], async_op=True) for s, e, _ in self.late_runs]
            for h in hs:
                h.wait()
        ev[4].record()
        self.gC.replay()
        ev[5].record()
        return self.loss, ev

    @staticmethod
    def phase_ms(ev):
        names = ("towers_forward", "all_gather", "loss_backward_pack", "all_reduce", "adam")
        return {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)}

    def replay(self):
        self.gA.replay()
        all_gather_rows(self.full, self.packed)
        self.gB.replay()
        if self.split is None:
            all_reduce_sum(self.flat)
        else:
            hs = [all_reduce_sum(self.flat[s:e], async_op=True) for s, e, _ in self.early_runs]
            self.gB2.replay()                                           # runs while the early ranges are being reduced
            hs += [all_reduce_sum(self.flat[s:e], async_op=True) for s, e, _ in self.late_runs]
            for h in hs:
                h.wait()
        self.gC.replay()
        return self.loss
