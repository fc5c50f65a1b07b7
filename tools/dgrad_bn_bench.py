#!/usr/bin/env python
"""A/B of the data gradient that takes the BatchNorm-backward sums in its epilogue (tri_conv_dgrad_bn) against data gradient +
tri_bn_bwd_reduce, one ResNet layer geometry at a time (graph-captured, GPU time per call).
python tools/dgrad_bn_bench.py [--precision f16] [--batch 32] [--views 6] [--image 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tricolo_amd import ops  # noqa: E402
from tools.conv_layers_bench import time_it  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--views", type=int, default=6)
    ap.add_argument("--image", type=int, default=128)
    args = ap.parse_args()
    dev, prec = "cuda", args.precision
    store = torch.float16 if prec == "f16" else torch.bfloat16
    N, h, inpl = args.batch * args.views, args.image // 4, 64
    for planes, stride in ((64, 1), (128, 2), (256, 2), (512, 2)):
        h = h // stride
        g = ops.ConvGeom(N, (1, h, h), planes, planes, planes, (1, 3, 3), 1, (0, 1, 1), (planes * 9, 1, 9))
        w = torch.randn(planes, planes, 1, 3, 3, device=dev) * 0.05
        tp = ops.pack_weight(w, g, prec, transposed=True)
        shape = (N, 1, h, h, planes)
        dy = torch.randn(shape, device=dev).to(store)
        # the BatchNorm input / saved output are COLD in the step (written a forward ago): rotate over enough copies to defeat the
        # 256 MB infinity cache
        ncopy = max(2, int(700e6 // (2 * torch.empty(shape).numel() * 2)))
        ys = [torch.randn(shape, device=dev).to(store) for _ in range(ncopy)]
        ros = [torch.randn(shape, device=dev).to(store) for _ in range(ncopy)]
        y, ro = ys[0], ros[0]
        turn = [0]

        def nxt():
            turn[0] += 1
            return ys[turn[0] % ncopy], ros[turn[0] % ncopy]
        base = torch.randn(shape, device=dev).to(store)
        co = ops.BNCoeffs(planes, dev)
        co.scale.fill_(1.0); co.shift.fill_(0.1); co.mean.zero_(); co.invstd.fill_(1.0)
        gamma = torch.ones(planes, device=dev)
        M = N * h * h
        nblk = ops.lib().tri_bn_bwd_num_blocks(M)
        part = torch.empty((nblk, 2, planes), dtype=torch.float32, device=dev)

        def reduce1(gr):
            y, _ = nxt()
            ops.check(ops.lib().tri_bn_bwd_reduce(ops.ptr(y), ops.ptr(gr), M, planes, ops.ptr(part), ops.ptr(co.scale), ops.ptr(co.shift), None, None,
                                                  ops._abf(y), ops.stream()), "reduce")

        def reduce2(gr):
            y, ro = nxt()
            ops.check(ops.lib().tri_bn_bwd_reduce(ops.ptr(y), ops.ptr(gr), M, planes, ops.ptr(part), None, None, ops.ptr(ro), None, ops._abf(y),
                                                  ops.stream()), "reduce")
        out = torch.empty(shape, dtype=store, device=dev)
        t_d = time_it(lambda: ops.conv_dgrad(dy, g, tp, out=out))
        t_dr = time_it(lambda: reduce1(ops.conv_dgrad(dy, g, tp, out=out)))
        fused = ops.conv_dgrad(dy, g, tp, out=out, bn_sums=(y, co, None))[1] is not None
        t_f = time_it(lambda: ops.conv_dgrad(dy, g, tp, out=out, bn_sums=(nxt()[0], co, None)))
        t_a = time_it(lambda: ops.conv_dgrad(dy, g, tp, out=base, accumulate=True))
        t_ar = time_it(lambda: reduce2(ops.conv_dgrad(dy, g, tp, out=base, accumulate=True)))
        t_af = time_it(lambda: ops.conv_dgrad(dy, g, tp, out=base, accumulate=True, bn_sums=(nxt()[0], None, nxt()[1])))
        print(f"l{planes} {h}x{h} x{N} fused={fused}: dgrad {t_d*1e3:6.1f} us, + reduce {t_dr*1e3:6.1f}, fused {t_f*1e3:6.1f} | "
              f"accumulate {t_a*1e3:6.1f}, + reduce {t_ar*1e3:6.1f}, fused {t_af*1e3:6.1f}", flush=True)


if __name__ == "__main__":
    main()
