"""SparseCNNEncoder on hand-written gfx950 kernels - drop-in for
/root/reference/tricolo/model/module/voxel_encoder/sparse_cnn.py:8-51.

Same constructor kwargs (voxel_size, ef_dim, z_dim, out_dim, **kwargs), same forward(x={'locs','feats'}, batch_size)
-> [B, out_dim] unit rows, same state-dict keys (sparseModel.{0,4,8,12,16}.weight in spconv's [Cout,kd,kh,kw,Cin]
layout, sparseModel.{1,5,9,13,17}.{weight,bias,running_mean,running_var,num_batches_tracked}, mlp.{0,2}.*).
One deliberate generalisation: mlp[0].in_features = z_dim * (voxel_size // 32)**3 instead of
the hard-coded 4096 (identical at 64^3, and makes the 32^3 configs of BASELINE.json runnable).

MI355X design: the COO batch is scattered once into a dense channels-last grid + site mask; per level the active sites are
listed in ascending order (tri_mask_compact) and every SubMConv3d / its data gradient is an implicit GEMM on MFMA over THAT
row list - executed work = active work, rows of inactive sites are neither computed nor written and no consumer reads them
(small deep levels run split-K over the list: slab row = list row).  In the 16-bit modes the two finest levels' forwards run the
BRICK kernels of csrc/conv_vox.hip instead (dense grid + site mask, a brick + halo staged once in LDS, 16-site runs without an
active site skipped); BatchNorm statistics come out of the conv epilogue; BN + ReLU + mask + 2^3 max-pool is one HBM pass.  Forward and backward
of the whole tower are ONE autograd node, so a step costs a handful of Python calls and is HIP-graph capturable.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

from .... import ops
from ....layers import TriModule, linear_bwd, linear_fwd, require_gpu


class SubMConv3dParams(nn.Module):
    """Parameter holder named like spconv.SubMConv3d(bias=False): weight [Cout, 3, 3, 3, Cin]."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, 3, 3, 3, cin))
        bound = 1.0 / math.sqrt(cin * 27)
        nn.init.uniform_(self.weight, -bound, bound)


class SparseCNNEncoder(TriModule):
    def __init__(self, voxel_size, ef_dim, z_dim, out_dim, precision=None, **kwargs):
        super().__init__()
        if voxel_size % 32 != 0:
            raise ValueError("voxel_size must be a multiple of 32 (five stride-2 pools)")
        self.voxel_size = voxel_size
        self.precision = precision
        self.chans = [3, ef_dim, ef_dim * 2, ef_dim * 4, ef_dim * 8, z_dim]
        mods = {}
        for i in range(5):                                  # sparse_cnn.py:12-35: conv at 4i, BN at 4i+1
            mods[str(4 * i)] = SubMConv3dParams(self.chans[i], self.chans[i + 1])
            mods[str(4 * i + 1)] = nn.BatchNorm1d(self.chans[i + 1])
        self.sparseModel = nn.ModuleDict(mods)
        self.spatial = (voxel_size // 32) ** 3
        self.mlp = nn.Sequential(nn.Linear(z_dim * self.spatial, out_dim), nn.ReLU(inplace=True), nn.Linear(out_dim, out_dim))
        self.fuse_pool_reduce = True                            # see _backward_impl (TRICOLO_POOL_REDUCE=0 / 1 overrides)
        self._geoms = {}
        self.__dict__["_packers"] = {}
        self.__dict__["_packed"] = {}

    # ------------------------------------------------------------------ parameter plumbing
    def _param_list(self):
        ps = []
        for i in range(5):
            bn = self.sparseModel[str(4 * i + 1)]
            ps += [self.sparseModel[str(4 * i)].weight, bn.weight, bn.bias]
        ps += [self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias]
        return ps

    def _geom(self, B, level):
        key = (B, level)
        g = self._geoms.get(key)
        if g is None:
            D = self.voxel_size >> level
            cin, cout = self.chans[level], self.chans[level + 1]
            cs = 4 if cin == 3 else cin
            g = ops.ConvGeom(B, (D, D, D), cin, cs, cout, (3, 3, 3), 1, (1, 1, 1), (27 * cin, cin, 1))
            self._geoms[key] = g
        return g

    def _prec(self):
        return self.precision or ops.default_precision()

    def _pack_all(self, B, prec, train, device):
        """One packing launch for the five conv layers (forward operands + data-gradient operands of levels 1-4)."""
        key = (B, train)
        packer = self._packers.get(key)
        if packer is None:
            packer = ops.WeightPacker()
            for l in range(5):
                w, g = self.sparseModel[str(4 * l)].weight, self._geom(B, l)
                packer.add((l, False), w, g)
                if train and l > 0:
                    packer.add((l, True), w, g, transposed=True)
            self._packers[key] = packer
        return packer.run(prec, device)

    # ------------------------------------------------------------------ forward / backward implementations
    def _forward_impl(self, locs, feats, B, save: bool):
        prec, V, train = self._prec(), self.voxel_size, self.training
        if ops.TIMELINE is not None and ops.TIMELINE.get("fine"):
            ops.stamp("voxel.fwd.start")
        self._packed = self._pack_all(B, prec, train and save, feats.device)
        if ops.TIMELINE is not None and ops.TIMELINE.get("fine"):
            ops.stamp("voxel.fwd.packed")
        if locs is None:                                        # feats = dense RGBA u8 grids [B,4,V,V,V] (SURVEY 8f-2 input)
            x, mask = ops.voxel_from_rgba(feats, dtype=ops.act_dtype(prec))
        else:
            x, mask = ops.voxel_scatter(locs, feats, B, V, dtype=ops.act_dtype(prec))
        compact = os.environ.get("TRICOLO_VOXEL_COMPACT", "1") != "0"        # A/B switch: mask-only (tile skipping) path
        fine = ops.TIMELINE is not None and ops.TIMELINE.get("fine")         # tools/step_timeline.py with TRICOLO_FINE_STAMPS=1
        if fine:
            ops.stamp("voxel.fwd.scattered")
        # (round 5) every level's site mask and row list up front, three launches (ops.mask_pyramid / mask_compact_multi) instead of two per
        # level between the convolutions; V % 16 != 0: level by level as before
        pyramid = V % 16 == 0
        if pyramid:
            masks = [mask] + ops.mask_pyramid(mask, B, V)
            lists = ops.mask_compact_multi(masks, [B * (V >> l) ** 3 for l in range(5)])
            rows = lists[0]
        else:
            rows = ops.mask_compact(mask, B * V ** 3)                         # (active positions ascending, device count)
        count = rows[1]
        saved = {"levels": [], "B": B}
        for l in range(5):
            D, C = V >> l, self.chans[l + 1]
            g = self._geom(B, l)
            conv, bn = self.sparseModel[str(4 * l)], self.sparseModel[str(4 * l + 1)]
            packed = self._packed[(l, False)]
            mode = ops._conv_mode(x, packed[1])
            use_rows = compact and not g.brick(False, mode)      # brick kernels walk the grid by the mask; split-K levels take the list too
            sel = dict(rows=rows) if use_rows else dict(row_mask=mask)
            if train:
                y, stats = ops.conv_fwd(x, g, packed, want_stats=True, **sel)
                co = ops.bn_finalize(stats, C, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                     count_dev=count, momentum=bn.momentum, eps=bn.eps)
            else:
                y = ops.conv_fwd(x, g, packed, **sel)
                co = ops.bn_eval_coeffs(C, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
            if fine:
                ops.stamp(f"voxel.fwd.l{l}.conv+stats")
            if pyramid:
                pooled, _ = ops.bn_relu_pool3d_fwd(y, co, mask, B, D, C, want_mask=False)
                mask_out, rows_out = (masks[l + 1], lists[l + 1]) if l < 4 else (None, (None, None))
            else:
                pooled, mask_out = ops.bn_relu_pool3d_fwd(y, co, mask, B, D, C)
                rows_out = ops.mask_compact(mask_out, B * (D // 2) ** 3) if l < 4 else (None, None)
            if fine:
                ops.stamp(f"voxel.fwd.l{l}.pooled")
            if save:
                saved["levels"].append((x, y, mask, count, co, pooled, rows, use_rows))
            x, mask, rows, count = pooled, mask_out, rows_out, rows_out[1]
        flat = ops.cast_to_f32(x.view(B, -1))                  # channels-last [B, v, v, v, C]; the head runs in fp32
        h = linear_fwd(flat, self.mlp[0].weight, self.mlp[0].bias, 1, prec, spatial=self.spatial)
        o = linear_fwd(h, self.mlp[2].weight, self.mlp[2].bias, 0, prec)
        z, norm = ops.l2norm_fwd(o)
        if save:
            saved.update(flat=flat, h=h, o=o, z=z, norm=norm)
        return z, saved

    def _backward_impl(self, saved, dz):
        prec, V, B = self._prec(), self.voxel_size, saved["B"]
        grads = [None] * 19
        ops.stamp("voxel.bwd.start")
        do = ops.l2norm_bwd(saved["z"], saved["norm"], dz)
        dh, grads[17], grads[18] = linear_bwd(saved["h"], self.mlp[2].weight, saved["o"], do, 0, prec)
        dflat, grads[15], grads[16] = linear_bwd(saved["flat"], self.mlp[0].weight, saved["h"], dh, 1, prec, spatial=self.spatial)
        gs = ops.grad_scale(prec)                              # f16 mode: activation gradients carried times gs (see mv_cnn.py)
        ugs = 1.0 / gs
        dx = ops.cast_from_f32(dflat.contiguous(), ops.act_dtype(prec), gs)
        batch = ops.wgrad_batch(dz.device)                     # the five weight-gradient reduces in one launch at the end
        compact = os.environ.get("TRICOLO_VOXEL_COMPACT", "1") != "0"
        # route + BatchNorm-backward sums in one launch (pool3d_bwd_route_reduce): Bi(V) 1.314 -> 1.302 ms.  Rounds 2-3 had TriCoLoNet switch
        # it off beside an image tower at 32^3 (the shorter voxel chain made the replayed graph fold its branches differently: 3.30 -> 3.33 ms);
        # since round 4 it stays on everywhere (tricolo_net.py).  The fused form calls finalize + apply itself, so the one-launch
        # tri_bn_bwd_small (tensors of <= 512 rows) is reached through ops.bn_bwd only - by the image tower's smallest layers and by this
        # tower with TRICOLO_POOL_REDUCE=0.
        env = os.environ.get("TRICOLO_POOL_REDUCE")
        fuse = self.fuse_pool_reduce if env is None else env == "1"
        for l in range(4, -1, -1):
            D, C = V >> l, self.chans[l + 1]
            g = self._geom(B, l)
            x, y, mask, count, co, pooled, rows, _ = saved["levels"][l]
            conv, bn = self.sparseModel[str(4 * l)], self.sparseModel[str(4 * l + 1)]
            # level 0 has no data gradient that would gather dy at inactive sites; its weight gradient walks a row list (compact) or is
            # the brick kernel, which masks dOut rows itself - only the masked-tile form of conv_wgrad_kernel needs the zeros.  The brick
            # kernel exists for 16-bit storage only (g.wgrad_brick is geometry-only): the SAME predicate picks the kernel and the rows
            # left unwritten, or fp32 storage with TRICOLO_VOXEL_COMPACT=0 would contract over uninitialised dy rows
            brick_wgrad = g.wgrad_brick and y.dtype != torch.float32
            rows_out = saved["levels"][l + 1][6] if l < 4 else None    # the pooled level's active-site list
            dy, dgamma, dbeta = ops.pool3d_bn_bwd(y, co, mask, pooled, dx.contiguous(), B, D, C, bn.weight, count, out_scale=ugs,
                                                  fused=fuse, keep_inactive=(l == 0 and (compact or brick_wgrad)), rows=rows, rows_out=rows_out)
            if compact and not brick_wgrad:
                # contraction over the active sites only (row list of the level)
                grads[3 * l] = ops.conv_wgrad(x, dy, g, conv.weight, prec, rows=rows, out_scale=ugs, batch=batch)
            else:
                grads[3 * l] = ops.conv_wgrad(x, dy, g, conv.weight, prec, row_mask=mask, out_scale=ugs, batch=batch)
            grads[3 * l + 1], grads[3 * l + 2] = dgamma, dbeta
            if ops.TIMELINE is not None and ops.TIMELINE.get("fine"):
                ops.stamp(f"voxel.bwd.l{l}.bn+wgrad_issue")
            if l > 0:
                pt = self._packed[(l, True)]
                if compact and not g.brick(True, ops._conv_mode(dy, pt[1])):
                    dx = ops.conv_dgrad(dy, g, pt, rows=rows)                   # only the active input sites are computed / written
                else:                                                          # (conv_voxg_kernel walks the coarse grids by the site mask)
                    dx = ops.conv_dgrad(dy, g, pt, row_mask=mask)
        if batch is not None:
            batch.flush()
        ops.stamp("voxel.bwd.end")
        return grads

    def forward(self, x, batch_size):
        """x = {'locs', 'feats'} as data_module.py:52-64 builds it, or - beyond the reference - {'rgba': u8 [B,4,V,V,V]},
        the dataset's dense grids: active sites / features are then derived on the device (no CPU COO build)."""
        if "rgba" in x:
            locs, feats = None, x["rgba"]
            if tuple(feats.shape[1:]) != (4, self.voxel_size, self.voxel_size, self.voxel_size) or feats.dtype != torch.uint8:
                raise RuntimeError("voxels['rgba'] must be uint8 [B, 4, V, V, V]")
        else:
            locs, feats = x["locs"], x["feats"]
        require_gpu(feats, "SparseCNNEncoder")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _VoxelTowerFn.apply(self, locs, feats, int(batch_size), *self._param_list())
        z, _ = self._forward_impl(locs, feats, int(batch_size), save=False)
        return z


class _VoxelTowerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, locs, feats, batch_size, *params):
        z, saved = module._forward_impl(locs, feats, batch_size, save=True)
        ctx.module, ctx.saved = module, saved
        return z

    @staticmethod
    def backward(ctx, dz):
        grads = ctx.module._backward_impl(ctx.saved, dz.contiguous())
        ctx.saved = None
        return (None, None, None, None, *grads)
