"""MVCNNEncoder (ResNet-18 trunk + per-shape view max-pool) on hand-written gfx950 kernels - drop-in for
/root/reference/tricolo/model/module/img_encoder/mv_cnn.py:13-33 (resnet18 branch of SVCNN, :40-45).

Same constructor kwargs (z_dim, out_dim, cnn_name, num_views, **kwargs; `clip_model` is accepted and ignored like
the reference does), same forward(x=[B*Nv,3,S,S], data_dict) -> [B, out_dim] unit rows, and the same 126
state-dict keys: net_1.{0,1,4,5,6,7}.* (Sequential over the torchvision children conv1, bn1, relu, maxpool,
layer1..4, avgpool), net_2.{weight,bias}, mlp.{0,2}.{weight,bias}.  resnet34/50/efficientnet branches
(mv_cnn.py:46-59) are out of scope (no BASELINE config uses them) and raise.  ImageNet weights cannot be downloaded
offline: parameters start from torchvision's init rule; load a checkpoint through load_state_dict.

MI355X design: images are converted once to channels-last [N,H,W,4]; all 20 convs run as implicit-GEMM on MFMA with
BatchNorm statistics produced by the conv epilogue; BN(+residual)+ReLU is one pass; global-average-pool and the
view max are fused.  The whole tower (forward and backward) is one autograd node.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .... import ops
from ....layers import SideStream, TriModule, linear_bwd, linear_fwd, require_gpu


_DS_FWD = int(os.environ.get("TRICOLO_DS_FWD", "2"))           # where the shortcut branch of layer2-4's first block is issued (round 6; see _run_block)
_DS_BWD = int(os.environ.get("TRICOLO_DS_BWD", "2"))
_PREP_DGRAD_LATE = os.environ.get("TRICOLO_PREP_DGRAD_LATE", "1") != "0"   # trunk data-gradient operands packed behind layer4 (0: with the forward operands)
_BN_PAIR = True               # (round 6; module flag, no environment switch) one set of BatchNorm-backward passes for bn2 + the shortcut's BatchNorm
_STEM_BESIDE_WGRAD = os.environ.get("TRICOLO_STEM_BESIDE_WGRAD", "1") != "0"   # the stem's backward beside the tower's weight-gradient launches (0: in front of them)
_PREP_ISSUE = int(os.environ.get("TRICOLO_PREP_ISSUE", "1"))    # where the trunk's operand packing is issued: 0 first thing, 1 behind the stem conv, 2 behind the max-pool


class _BasicBlockParams(nn.Module):
    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        self.stride = stride


def _resnet18_trunk_params():
    layers = [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1)]
    inpl = 64
    for planes, stride in ((64, 1), (128, 2), (256, 2), (512, 2)):
        layers.append(nn.Sequential(_BasicBlockParams(inpl, planes, stride), _BasicBlockParams(planes, planes, 1)))
        inpl = planes
    layers.append(nn.AdaptiveAvgPool2d((1, 1)))
    net = nn.Sequential(*layers)
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return net


class MVCNNEncoder(TriModule):
    def __init__(self, z_dim, out_dim, cnn_name, num_views, precision=None, **kwargs):
        super().__init__()
        if cnn_name != "resnet18":
            raise NotImplementedError(f"cnn_name={cnn_name!r}: only the resnet18 branch (mv_cnn.py:43-45) is built")
        self.num_views = num_views
        self.precision = precision
        self.net_1 = _resnet18_trunk_params()
        self.net_2 = nn.Linear(512, z_dim)
        self.mlp = nn.Sequential(nn.Linear(z_dim, out_dim), nn.ReLU(inplace=True), nn.Linear(out_dim, out_dim))
        self._geoms = {}
        self.__dict__["_packers"] = {}
        self.__dict__["_packed"] = {}
        self.__dict__["_side"] = SideStream("img")
        self.__dict__["_side_ds"] = SideStream("imgds")         # down-sample branch of layer2-4's first block
        self.__dict__["_side_prep"] = SideStream("imgprep")     # operand packing of layer1-4 next to the stem
        self.__dict__["split"] = None                           # parallel.BackwardSplit: lower / upper halves as separate autograd nodes

    def _prec(self):
        return self.precision or ops.default_precision()

    # conv/bn units in execution order; each = (conv_module, bn_module)
    def _blocks(self):
        return [blk for li in (4, 5, 6, 7) for blk in self.net_1[li]]

    def _param_list(self):
        ps = [self.net_1[0].weight, self.net_1[1].weight, self.net_1[1].bias]
        for blk in self._blocks():
            ps += [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias]
            if blk.downsample is not None:
                ps += [blk.downsample[0].weight, blk.downsample[1].weight, blk.downsample[1].bias]
        ps += [self.net_2.weight, self.net_2.bias, self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias]
        return ps

    def _geom2d(self, N, H, W, conv: nn.Conv2d):
        key = (N, H, W, id(conv))
        g = self._geoms.get(key)
        if g is None:
            cin, cout = conv.in_channels, conv.out_channels
            k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            cs = 4 if cin == 3 else cin
            g = ops.ConvGeom(N, (1, H, W), cin, cs, cout, (1, k, k), s, (0, p, p), (cin * k * k, 1, k * k))
            self._geoms[key] = g
        return g

    def _pack_all(self, N, H, W, prec, train, device):
        """Two packing launches: the stem's operand rows inline (the stem conv needs them at once), every other conv of the trunk
        (forward operands, plus data-gradient operands when training: 45 MB read + written, ~50 us) on a side stream NEXT TO the
        stem conv / BatchNorm / max-pool, which it used to precede on the tower's critical path.  join_packing() before layer1."""
        key = (N, H, W, train)
        packer = self._packers.get(key)
        if packer is None:
            packer = (ops.WeightPacker(), ops.WeightPacker(), ops.WeightPacker())     # stem | trunk forward | trunk data-gradient operands
            stem_g = self._geom2d(N, H, W, self.net_1[0])
            packer[0].add((id(self.net_1[0]), False), self.net_1[0].weight, stem_g)
            convs = []
            h, w = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
            h, w = (h + 1) // 2, (w + 1) // 2
            for blk in self._blocks():
                s = blk.conv1.stride[0]
                convs.append((blk.conv1, h, w))
                if blk.downsample is not None:
                    convs.append((blk.downsample[0], h, w))
                h, w = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
                convs.append((blk.conv2, h, w))
            for conv, ch, cw in convs:
                g = self._geom2d(N, ch, cw, conv)
                packer[1].add((id(conv), False), conv.weight, g)
                if train:
                    packer[2 if _PREP_DGRAD_LATE else 1].add((id(conv), True), conv.weight, g, transposed=True)
            self._packers[key] = packer
        packed = dict(packer[0].run(prec, device))
        ev = None
        if _PREP_ISSUE:                                        # (the trunk's packing depends on nothing issued in this step ...)
            ev = torch.cuda.Event()
            ev.record()

        def pack_rest():
            with torch.cuda.stream(self._side_prep.fork(event=ev)):
                rest = packer[1].run(prec, device)
            self._packed.update(rest)
            self.__dict__["_prep_pending"] = [t for pair in rest.values() for t in pair if t is not None]

        def pack_dgrad():
            # Round 6 (_PREP_DGRAD_LATE): the data-gradient operands (half of the trunk's packing traffic) are not needed before the
            # backward: they are packed behind layer4's last launch, in the shadow of the heads / loss section (a chain of tiny kernels with
            # the GPU nearly idle), instead of beside the stem conv, which is HBM-bound like the packing itself (83 us in-graph against 48 alone)
            with torch.cuda.stream(self._side_prep.fork(event=ev)):
                rest = packer[2].run(prec, device)
            self._packed.update(rest)
            self.__dict__["_prep_pending_dgrad"] = [t for pair in rest.values() for t in pair if t is not None]
        self.__dict__["_pack_dgrad"] = pack_dgrad if (train and _PREP_DGRAD_LATE and packer[2].entries) else None
        if ev is None and self.__dict__["_pack_dgrad"] is not None:
            ev = torch.cuda.Event()
            ev.record()
        # Round 6 (_PREP_ISSUE = 1): ... but is ISSUED behind the stem conv's launch: a replayed graph releases nodes in capture order, and the
        # 67 us packing launch captured first took the CUs the stem conv wanted (79 us in-graph against 48 alone)
        self.__dict__["_pack_rest"] = pack_rest
        self._packed = packed
        if not _PREP_ISSUE:
            self._run_pack_rest()
        return packed

    def _run_pack_rest(self):
        fn = self.__dict__.get("_pack_rest")
        if fn is not None:
            self.__dict__["_pack_rest"] = None
            fn()

    def _run_pack_dgrad(self):
        fn = self.__dict__.get("_pack_dgrad")
        if fn is not None:
            self.__dict__["_pack_dgrad"] = None
            fn()

    def _join_packing_dgrad(self):
        self._run_pack_dgrad()                                  # (a backward without the upper half's forward in between: never skipped)
        pend = self.__dict__.get("_prep_pending_dgrad")
        if pend:
            self._side_prep.join(*pend)
            self.__dict__["_prep_pending_dgrad"] = None

    def _join_packing(self):
        """The trunk's operand rows (packed on the side stream by _pack_all) are needed from here on."""
        pend = self.__dict__.get("_prep_pending")
        if pend:
            self._side_prep.join(*pend)
            self.__dict__["_prep_pending"] = None

    def _conv_bn(self, x, conv, bn, prec, train, after_conv=None):
        N, _, H, W, _ = x.shape
        g = self._geom2d(N, H, W, conv)
        packed = self._packed[(id(conv), False)]
        if train:
            y, stats = ops.conv_fwd(x, g, packed, want_stats=True)
            if after_conv is not None:
                after_conv()                                      # (issued between the conv launch and its BatchNorm finalize)
            co = ops.bn_finalize(stats, g.cout, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                 count_host=g.M, momentum=bn.momentum, eps=bn.eps)
        else:
            y = ops.conv_fwd(x, g, packed)
            co = ops.bn_eval_coeffs(g.cout, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        return y, co, g

    # The tower is written as a LOWER half (stem, layer1, layer2 -> feature map x2) and an UPPER half (layer3, layer4, pool,
    # heads).  Single-GPU steps run both inside ONE autograd node; the data-parallel step installs a parallel.BackwardSplit
    # (self.split) and gets two nodes with the split's gradient gate between them: 96 % of the tower's gradient bytes (layer3-4
    # + heads) are then final when the upper node's backward returns and are all-reduced under the lower half's backward
    # (parallel.dp_training_step).
    N_LOWER_BLOCKS = 4

    def _lower_params(self):
        return self._param_list()[:3 + sum(9 if b.downsample is not None else 6 for b in self._blocks()[:self.N_LOWER_BLOCKS])]

    def _upper_params(self):
        return self._param_list()[len(self._lower_params()):]

    def _run_block(self, blk, x, prec, train, save, store):
        ds = blk.downsample is not None
        mode = _DS_FWD if (ds and train) else 0
        shortcut = []
        ev = None
        if mode == 2:                                          # (the shortcut depends on the block input only ...)
            ev = torch.cuda.Event()
            ev.record()

        def run_shortcut():                                    # 1x1/2 conv + BN of the shortcut: independent of conv1 / conv2
            with torch.cuda.stream(self._side_ds.fork(x, event=ev)):
                shortcut.extend(self._conv_bn(x, blk.downsample[0], blk.downsample[1], prec, train))
        # Round 6: where the shortcut's side branch is ISSUED matters under HIP-graph replay - nodes are released in capture order, so a branch
        # captured first gets the CUs first and conv1, whose launch wants whole CUs (LDS, registers), runs part of its workgroups in a second
        # round (in-graph trace: conv1 31.7 / 38.5 us beside the shortcut conv against 19-23 us alone).  _DS_FWD: 0 = forked and issued at the
        # start of the block; 1 = forked BEHIND conv1's launch (it waits for conv1 and runs beside conv1's BatchNorm passes); 2 = issued
        # behind conv1's launch but depending only on the block input (... and is released after conv1, taking the CUs conv1 leaves)
        if ds and not mode:
            run_shortcut()
        fine = ops.TIMELINE is not None and ops.TIMELINE.get("fine")
        y1, co1, g1 = self._conv_bn(x, blk.conv1, blk.bn1, prec, train, after_conv=run_shortcut if mode else None)
        if ds:
            yd, cod, gd = shortcut
        if fine:
            ops.stamp(f"image.fwd.c{y1.shape[-1]}.conv1+fin")
        a1 = ops.bn_act(y1, co1, relu=True)
        if fine:
            ops.stamp(f"image.fwd.c{y1.shape[-1]}.act1")
        y2, co2, g2 = self._conv_bn(a1, blk.conv2, blk.bn2, prec, train)
        if fine:
            ops.stamp(f"image.fwd.c{y1.shape[-1]}.conv2+fin")
        if ds:
            self._side_ds.join(yd, cod.scale, cod.shift)
            out = ops.bn_act(y2, co2, relu=True, res=yd, res_co=cod)
        else:
            yd = cod = gd = None
            out = ops.bn_act(y2, co2, relu=True, res=x)
        if fine:
            ops.stamp(f"image.fwd.c{y1.shape[-1]}.act2")
        if save:
            store.append((x, y1, co1, g1, a1, y2, co2, g2, yd, cod, gd, out))
        return out

    def _forward_lower(self, images, save: bool):
        prec, train = self._prec(), self.training
        N = images.shape[0]
        if N % self.num_views:
            raise RuntimeError("mat shape: number of images is not a multiple of num_views")
        self._packed = self._pack_all(N, images.shape[2], images.shape[3], prec, train and save, images.device)
        if images.dtype == torch.uint8:                         # raw renderings: ToTensor + CLIP Normalize on the device (8f-2)
            x0 = ops.nchw3_u8_to_nhwc4(images, dtype=ops.act_dtype(prec))
        else:
            x0 = ops.nchw3_to_nhwc4(images, dtype=ops.act_dtype(prec))
        def after_stem_conv():
            if _PREP_ISSUE == 1:
                self._run_pack_rest()
        y, co, g = self._conv_bn(x0, self.net_1[0], self.net_1[1], prec, train, after_conv=after_stem_conv)
        x, parg = ops.maxpool2d_fwd(y, want_arg=save, bn=co)            # BN + ReLU + 3x3/2 max-pool: relu(bn(y)) is never stored
        self._run_pack_rest()                                          # (_PREP_ISSUE = 2, or an eval-mode forward: after_conv is a training hook)
        ops.stamp("image.fwd.stem.end")
        saved = {"stem": (x0, y, co, g, parg), "blocks": [], "N": N}
        self._join_packing()
        for blk in self._blocks()[:self.N_LOWER_BLOCKS]:
            x = self._run_block(blk, x, prec, train, save, saved["blocks"])
        ops.stamp("image.fwd.layer2.end")
        return x, saved

    def _forward_upper(self, x, save: bool):
        prec, train = self._prec(), self.training
        N = x.shape[0]
        B = N // self.num_views
        saved = {"blocks": [], "B": B, "N": N}
        for blk in self._blocks()[self.N_LOWER_BLOCKS:]:
            x = self._run_block(blk, x, prec, train, save, saved["blocks"])
        ops.stamp("image.fwd.layer4.end")
        if save:
            self._run_pack_dgrad()
        pooled, arg = ops.avgpool_viewmax_fwd(x, B, self.num_views)
        f = linear_fwd(pooled, self.net_2.weight, self.net_2.bias, 0, prec)
        h = linear_fwd(f, self.mlp[0].weight, self.mlp[0].bias, 1, prec)
        o = linear_fwd(h, self.mlp[2].weight, self.mlp[2].bias, 0, prec)
        zz, norm = ops.l2norm_fwd(o)
        if save:
            self._join_packing_dgrad()      # (behind the heads' launches: a forward captured on its own - parallel.GraphedDPStep - must end joined)
            saved.update(feat_shape=tuple(x.shape), pooled=pooled, arg=arg, f=f, h=h, o=o, z=zz, norm=norm)
        return zz, saved

    def _forward_impl(self, images, save: bool):
        x2, lo = self._forward_lower(images, save)
        zz, hi = self._forward_upper(x2, save)
        return zz, {"lower": lo, "upper": hi}

    def _backward_blocks(self, blocks, saved_blocks, dout, gr, prec, ugs, batch=None):
        """BasicBlock backward over `blocks` (last first); returns the gradient w.r.t. the first block's input."""
        side = self._side
        # Weight gradients need only x and dy and could run beside the dgrad / BatchNorm-backward chain on a side stream
        # (pattern "s"; "sm" alternates).  Measured with the round-1 kernels that is a LOSS: two GPU-filling kernels
        # side by side thrash each other (all on the side stream 4.23 ms per step, alternating 4.09-4.15, all inline 3.93-3.97),
        # so they are issued inline on the tower's stream.
        pattern = "m"                                              # (the side-stream pattern experiment switch was dropped in round 6)
        turn = [0]

        def wgrad_async(x, dy, g, w):
            ch = pattern[turn[0] % len(pattern)]
            turn[0] += 1
            if ch == "m":
                gr[w] = ops.conv_wgrad(x, dy, g, w, prec, out_scale=ugs, batch=batch)     # reduced by the caller's batch.flush()
                return
            with torch.cuda.stream(side.fork(x, dy)):
                gr[w] = ops.conv_wgrad(x, dy, g, w, prec, out_scale=ugs)

        dout_sums = None                       # BatchNorm-backward sums of bn2 taken by the data gradient that produced dout
        for bi in range(len(blocks) - 1, -1, -1):
            blk, sv = blocks[bi], saved_blocks[bi]
            x, y1, co1, g1, a1, y2, co2, g2, yd, cod, gd, out = sv
            # relu(bn2(y2) + residual) backward inside the BN passes; g = dout * (out > 0) (gradient of the pre-activation sum,
            # also the residual branch's gradient) is written by the apply pass in place of dout
            # (round 6) a down-sampling block's two BatchNorms - bn2 and the shortcut's - receive the same gradient: one reduce / finalize / apply
            # for both (ops.bn_bwd_pair: g and the saved output read once, six launches less per step, bit-equal to the two single calls); the side
            # branch below is left with the shortcut's 1x1 / 2 data gradient.  Step: -19 / -4 / +10 us in three alternating A/Bs on three boxes.
            pair = (_BN_PAIR and blk.downsample is not None and batch is not None and dout_sums is None and yd is not None
                    and yd.shape == y2.shape and ops._sync_world() == 1)
            dyd_pair = None
            if pair:
                (dy2, gr[blk.bn2.weight], gr[blk.bn2.bias], dyd_pair, gr[blk.downsample[1].weight],
                 gr[blk.downsample[1].bias]) = ops.bn_bwd_pair(y2, co2, blk.bn2.weight, yd, cod, blk.downsample[1].weight, dout, out, g2.M,
                                                               g_masked=dout, out_scale=ugs)
            else:
                dy2, gr[blk.bn2.weight], gr[blk.bn2.bias] = ops.bn_bwd(y2, dout, co2, blk.bn2.weight, count_host=g2.M, inplace=False,
                                                                       relu_out=out, g_masked=dout, out_scale=ugs, partial=dout_sums)
            g = dout
            fine = ops.TIMELINE is not None and ops.TIMELINE.get("fine")
            tag = f"image.bwd.c{y2.shape[-1]}.b{bi}"
            if fine:
                ops.stamp(tag + ".bn2")
            # shortcut branch (three BatchNorm passes + the 1x1 / 2 data gradient) next to the conv2 / conv1 chain; _DS_BWD as _DS_FWD in
            # _run_block: 0 = forked and issued here, 1 = forked behind conv2's data gradient (runs beside bn1's passes), 2 = issued behind
            # conv2's data gradient, depending on bn2's backward only
            mode = _DS_BWD if (blk.downsample is not None and batch is not None) else 0
            ev = None
            if mode == 2:
                ev = torch.cuda.Event()
                ev.record()

            def shortcut_bwd():
                with torch.cuda.stream(self._side_ds.fork(g, yd, event=ev)):
                    if dyd_pair is not None:
                        return dyd_pair, ops.conv_dgrad(dyd_pair, gd, self._packed[(id(blk.downsample[0]), True)])
                    dyd_, gr[blk.downsample[1].weight], gr[blk.downsample[1].bias] = ops.bn_bwd(
                        yd, g, cod, blk.downsample[1].weight, count_host=gd.M, inplace=False, out_scale=ugs)
                    if batch is None:
                        gr[blk.downsample[0].weight] = ops.conv_wgrad(x, dyd_, gd, blk.downsample[0].weight, prec, out_scale=ugs)
                    return dyd_, ops.conv_dgrad(dyd_, gd, self._packed[(id(blk.downsample[0]), True)])
            if blk.downsample is not None and not mode:
                dyd, dx = shortcut_bwd()
            wgrad_async(a1, dy2, g2, blk.conv2.weight)
            # relu(bn1(y1)) backward: the ReLU mask is recomputed from y1 inside the BN passes (no relu_bwd pass over a1); where conv2's
            # data-gradient kernel can, it takes bn1's sums in its epilogue (no reduce pass over da1 / y1)
            da1, sums1 = ops.conv_dgrad(dy2, g2, self._packed[(id(blk.conv2), True)], bn_sums=(y1, co1, None))
            if mode:
                dyd, dx = shortcut_bwd()
            if fine:
                ops.stamp(tag + ".dgrad2")
            dy1, gr[blk.bn1.weight], gr[blk.bn1.bias] = ops.bn_bwd(y1, da1, co1, blk.bn1.weight, count_host=g1.M, relu=True, out_scale=ugs,
                                                                   partial=sums1)
            if fine:
                ops.stamp(tag + ".bn1")
            wgrad_async(x, dy1, g1, blk.conv1.weight)
            if blk.downsample is not None:
                if batch is None:
                    self._side_ds.join(dx, gr[blk.downsample[0].weight], gr[blk.downsample[1].weight], gr[blk.downsample[1].bias])
                else:                                                          # the shortcut's weight gradient joins the tower's job queue
                    self._side_ds.join(dx, dyd, gr[blk.downsample[1].weight], gr[blk.downsample[1].bias])
                    gr[blk.downsample[0].weight] = ops.conv_wgrad(x, dyd, gd, blk.downsample[0].weight, prec, out_scale=ugs, batch=batch)
            else:
                dx = g                                                         # identity branch
            # conv1's data gradient completes dx = the previous block's dout: that block's bn2 sums (mask: its saved output = this x)
            if bi > 0:
                prev = saved_blocks[bi - 1]
                dx, dout_sums = ops.conv_dgrad(dy1, g1, self._packed[(id(blk.conv1), True)], out=dx, accumulate=True,
                                               bn_sums=(prev[5], None, prev[11]))
            else:
                dx, dout_sums = ops.conv_dgrad(dy1, g1, self._packed[(id(blk.conv1), True)], out=dx, accumulate=True), None
            if fine:
                ops.stamp(tag + ".dgrad1")
            dout = dx
        side.join(*[v for v in gr.values() if v.dim() == 4])
        return dout

    def _backward_upper(self, saved, dz, batch=None):
        """-> (gradient w.r.t. the lower half's output x2 - carried times ops.grad_scale in the f16 mode -, upper parameter grads).
        Weight-gradient reduces of the main stream are deferred into `batch` (one grouped launch; the caller's, or an own one)."""
        prec, B = self._prec(), saved["B"]
        gr = {}
        ops.stamp("image.bwd.start")
        self._join_packing_dgrad()
        own = batch is None
        if own:
            batch = ops.wgrad_batch(dz.device)
        do = ops.l2norm_bwd(saved["z"], saved["norm"], dz)
        dh, gr[self.mlp[2].weight], gr[self.mlp[2].bias] = linear_bwd(saved["h"], self.mlp[2].weight, saved["o"], do, 0, prec)
        df, gr[self.mlp[0].weight], gr[self.mlp[0].bias] = linear_bwd(saved["f"], self.mlp[0].weight, saved["h"], dh, 1, prec)
        dp, gr[self.net_2.weight], gr[self.net_2.bias] = linear_bwd(saved["pooled"], self.net_2.weight, saved["f"], df, 0, prec)
        # f16 mode: activation gradients are carried times gs (ops.F16_GRAD_SCALE); parameter-gradient kernels undo it
        gs = ops.grad_scale(prec)
        dout = ops.avgpool_viewmax_bwd(dp, saved["arg"], saved["feat_shape"], B, self.num_views, dtype=ops.act_dtype(prec), scale=gs)
        ops.stamp("image.bwd.heads.end")
        dx2 = self._backward_blocks(self._blocks()[self.N_LOWER_BLOCKS:], saved["blocks"], dout, gr, prec, 1.0 / gs, batch)
        ops.stamp("image.bwd.layer3.end")
        if own and batch is not None:
            batch.flush()
        return dx2, [gr[p] for p in self._upper_params()]

    def _backward_lower(self, saved, dout, batch=None, stem_side=None):
        prec = self._prec()
        ugs = 1.0 / ops.grad_scale(prec)
        gr = {}
        self._join_packing_dgrad()
        own = batch is None
        if own:
            batch = ops.wgrad_batch(dout.device)
        dout = self._backward_blocks(self._blocks()[:self.N_LOWER_BLOCKS], saved["blocks"], dout, gr, prec, ugs, batch)
        x0, y, co, g0, parg = saved["stem"]
        ops.stamp("image.bwd.layer1.end")
        stem_mode = "2"                                                # (2: apply pass inside the weight-gradient kernel; the 0 / 1 A/B forms stay reachable for odd map sizes only)
        if y.shape[2] % 2 == 0 and y.shape[3] % 2 == 0 and stem_mode == "2":
            # ... and the BatchNorm-backward apply pass inside the weight-gradient kernel's staging: the stem's dy is never stored
            # Round 6 (stem_side, _STEM_BESIDE_WGRAD): the stem's backward - three HBM-bound launches, ~85 us - has nothing behind it but the
            # tower's weight-gradient launches, which are MFMA-bound and need nothing from it: those are issued first (WgradBatch.prelaunch)
            # and the stem chain runs on the side stream beside them, hanging on the event recorded here
            cm = None
            if stem_side is not None and _STEM_BESIDE_WGRAD and batch is not None:
                ev = torch.cuda.Event()
                ev.record()
                if batch.prelaunch():
                    cm = torch.cuda.stream(stem_side.fork(dout, event=ev))
            import contextlib
            with (cm if cm is not None else contextlib.nullcontext()):
                gr[self.net_1[0].weight], gr[self.net_1[1].weight], gr[self.net_1[1].bias] = ops.maxpool_bn_bwd_wgrad(
                    x0, y, parg, dout, co, self.net_1[1].weight, g0, self.net_1[0].weight, prec, out_scale=ugs, batch=batch,
                    pooled=saved["blocks"][0][0])                  # (the max-pool output = the first block's saved input)
            if cm is not None and not torch.cuda.is_current_stream_capturing():
                for t_ in (gr[self.net_1[0].weight], gr[self.net_1[1].weight], gr[self.net_1[1].bias]):
                    t_.record_stream(torch.cuda.current_stream())   # (allocated on the side stream, read by the optimizer on this one)
            if own and batch is not None:
                batch.flush()
            return [gr[p] for p in self._lower_params()]
        if y.shape[2] % 2 == 0 and y.shape[3] % 2 == 0 and stem_mode != "0":
            # BatchNorm backward straight from the pooled gradient and the winning-tap map: no max-pool backward pass
            dy, gr[self.net_1[1].weight], gr[self.net_1[1].bias] = ops.maxpool_bn_bwd(y, parg, dout, co, self.net_1[1].weight, out_scale=ugs,
                                                                                      pooled=saved["blocks"][0][0])
        else:
            dzs = ops.maxpool2d_bwd(parg, dout, tuple(y.shape))
            dy, gr[self.net_1[1].weight], gr[self.net_1[1].bias] = ops.bn_bwd(y, dzs, co, self.net_1[1].weight, count_host=g0.M, relu=True,
                                                                              out_scale=ugs)
        gr[self.net_1[0].weight] = ops.conv_wgrad(x0, dy, g0, self.net_1[0].weight, prec, out_scale=ugs, batch=batch)
        if own and batch is not None:
            batch.flush()
        return [gr[p] for p in self._lower_params()]

    def _backward_impl(self, saved, dz):
        batch = ops.wgrad_batch(dz.device)                         # one grouped reduce for the whole tower
        dx2, up = self._backward_upper(saved["upper"], dz, batch)
        # (round 4: launching layer3 / layer4's weight-gradient jobs here on a side stream, beside the lower half's data-gradient /
        #  BatchNorm chain instead of behind it, does not shorten the step - 2.766 against 2.747 ms, three alternating pairs: the
        #  kernel-row kernel's workgroups fill every CU's registers and the chain's kernels wait for them either way)
        lo = self._backward_lower(saved["lower"], dx2, batch, stem_side=self._side)
        if batch is not None:
            batch.flush(side=self._side)
        ops.stamp("image.bwd.end")
        return lo + up

    def forward(self, x, data_dict=None):
        require_gpu(x, "MVCNNEncoder")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if self.split is not None:                            # two autograd nodes with the split's gate between them
                x2 = _MVCNNLowerFn.apply(self, x, *self._lower_params())
                return _MVCNNUpperFn.apply(self, self.split.gate(x2), *self._upper_params())
            return _MVCNNTowerFn.apply(self, x, *self._param_list())
        z, _ = self._forward_impl(x, save=False)
        return z


class _MVCNNTowerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, images, *params):
        z, saved = module._forward_impl(images, save=True)
        ctx.module, ctx.saved = module, saved
        return z

    @staticmethod
    def backward(ctx, dz):
        grads = ctx.module._backward_impl(ctx.saved, dz.contiguous())
        ctx.saved = None
        return (None, None, *grads)


class _MVCNNLowerFn(torch.autograd.Function):
    """stem + layer1 + layer2 as their own autograd node (data-parallel gradient overlap, see MVCNNEncoder.split_backward)."""

    @staticmethod
    def forward(ctx, module, images, *params):
        x2, saved = module._forward_lower(images, save=True)
        ctx.module, ctx.saved, ctx.nparams = module, saved, len(params)
        ctx.set_materialize_grads(False)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        if dx2 is None:                                           # the gate above is deferring: this node runs in stage 2
            return (None, None) + (None,) * ctx.nparams
        grads = ctx.module._backward_lower(ctx.saved, dx2.contiguous())
        ctx.saved = None
        return (None, None, *grads)


class _MVCNNUpperFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x2, *params):
        z, saved = module._forward_upper(x2, save=True)
        ctx.module, ctx.saved = module, saved
        return z

    @staticmethod
    def backward(ctx, dz):
        dx2, grads = ctx.module._backward_upper(ctx.saved, dz.contiguous())
        ctx.saved = None
        return (None, dx2, *grads)
