"""Forced-routing float64 replay of the voxel tower (shared by tests/test_gpu_modules.py and tools/voxel_replay_debug.py): see
test_voxel_backward_replay_with_forced_routing.  TEST INFRASTRUCTURE (imports oracle/)."""
import os

import torch
import torch.nn.functional as F

from oracle import modules as om
from oracle import spconv_dense as sp
from oracle.recipe import fill_module
from tricolo_amd.data import synthetic as syn

DEV = "cuda"


def voxel_routes(saved, B, V):
    """Per level: R [B, C, D, D, D] in {0, 1} = 1 at the child the HIP backward routes the pooled gradient to - the first child in
    (d, h, w) scan order whose rounded post-ReLU value equals the pooled maximum and is > 0 (bn_pool.hip pool3d_bwd_route_kernel) -
    recomputed from the tensors that kernel reads; (routes, number of windows that needed the arg-max fallback)."""
    routes, unmatched = [], 0
    for l in range(5):
        x, y, mask, count, co, pooled, rows, _ = saved["levels"][l]
        D, C = V >> l, y.shape[-1]
        Do = D // 2
        act = mask[:B * D ** 3].view(B, D, D, D, 1) != 0
        zz = torch.relu(torch.addcmul(co.shift.double(), y.double(), co.scale.double())).float()       # fma(y, scale, shift) as the kernel forms it
        zz = torch.where(act, zz.to(y.dtype).float(), torch.zeros((), device=y.device))                # storage rounding; unwritten rows of inactive sites out
        zc = zz.view(B, Do, 2, Do, 2, Do, 2, C).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(B, Do, Do, Do, C, 8)
        pm = pooled.float().unsqueeze(-1)
        hit = (zc == pm) & (zc > 0)
        first = hit & (hit.int().cumsum(-1) == 1)
        none = (pm.squeeze(-1) > 0) & ~hit.any(-1)                                                    # (a last-bit fma difference: take the arg-max instead)
        unmatched += int(none.sum().item())
        if none.any():
            am = F.one_hot(zc.argmax(-1), 8).bool()
            first = torch.where(none.unsqueeze(-1), am, first)
        R = first.view(B, Do, Do, Do, C, 2, 2, 2).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(B, D, D, D, C)
        routes.append(R.permute(0, 4, 1, 2, 3).double().cpu())
    return routes, unmatched


def voxel_forced_replay(V, B, seed_off=1, module=None, want_grads=False, poison=False):
    """Runs the HIP SparseCNNEncoder (current default precision) forward + backward on the synthetic batch, replays the float64 oracle
    with the routing forced to the HIP forward's, and returns ({parameter name: relative L2 error of its gradient}, unmatched windows,
    max |z_hip - z_oracle|) (+ the two gradient dicts with want_grads)."""
    from tricolo_amd.model.module.voxel_encoder.sparse_cnn import SparseCNNEncoder
    batch = syn.make_batch(B, voxel_size=V, num_views=None, seed=syn.BASE_SEED + seed_off)
    up = torch.randn((B, 512), generator=torch.Generator().manual_seed(17))
    m = module
    if m is None:
        m = SparseCNNEncoder(V, 32, 512, 512)
        fill_module(m, prefix="voxel_encoder.")
        m = m.to(DEV)
    vox = {k: v.to(DEV) for k, v in batch["voxels"].items()}
    for p in m.parameters():
        p.grad = None
    if poison:                                                        # NaNs into the allocator's free blocks: rows a kernel leaves unwritten must never be read
        torch.empty((64 << 20,), dtype=torch.float32, device=DEV).fill_(float("nan"))
    z = m(vox, B)
    (z * up.to(DEV)).sum().backward()
    with torch.no_grad():
        _, saved = m._forward_impl(vox["locs"], vox["feats"], B, save=True)
    routes, unmatched = voxel_routes(saved, B, V)
    ref = om.SparseCNNRef(V, 32, 512, 512)
    fill_module(ref, prefix="voxel_encoder.")
    ref = ref.double()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t = sp.SparseConvTensor(batch["voxels"]["feats"].double(), batch["voxels"]["locs"], [V] * 3, B)
    xd, md = t.dense, t.mask
    for l in range(5):
        conv, bn = ref.sparseModel[4 * l], ref.sparseModel[4 * l + 1]
        yd = F.conv3d(xd, conv.weight.permute(0, 4, 1, 2, 3), padding=1) * md
        td = sp.masked_batchnorm(bn, sp.SparseConvTensor.from_dense(yd, md)).dense
        xd = F.avg_pool3d(td * routes[l], 2) * 8.0                    # the forced winner's value (post-ReLU: it is > 0), zero where none won
        md = F.max_pool3d(md, 2, 2)
    hmask = (saved["h"] > 0).double().cpu()                           # the head's own ReLU (mlp[1]) is routing too: a hidden unit within the 16-bit
    zr = F.normalize(ref.mlp[2](ref.mlp[0](xd.reshape(B, -1)) * hmask), dim=1)   # towers' forward error of zero flips and moves every gradient below it
    zdiff = float((z.detach().double().cpu() - zr.detach()).abs().max())
    (zr * up.double()).sum().backward()
    rg = dict(ref.named_parameters())
    table = {}
    for name, p in m.named_parameters():
        a = rg[name].grad
        table[name] = float((p.grad.detach().double().cpu() - a).norm() / a.norm().clamp_min(1e-300))
    if want_grads:
        return table, unmatched, zdiff, {n: p.grad.detach().double().cpu() for n, p in m.named_parameters()}, {n: p.grad for n, p in rg.items()}
    return table, unmatched, zdiff
