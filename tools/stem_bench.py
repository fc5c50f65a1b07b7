#!/usr/bin/env python
"""The ResNet stem's HBM-bound passes in isolation, cache-cold: every call works on its own copy of the tensors (six copies of the
100 MB conv output rotate, so nothing is re-read out of the 256 MB Infinity Cache), HIP events around 12 calls, best of 3.
    python tools/stem_bench.py [images=192]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tricolo_amd import ops  # noqa: E402


def timeit(fn, n):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    dev = torch.device("cuda:0")
    C, H, W, R = 64, 64, 64, 6
    g = torch.Generator(device=dev).manual_seed(1)
    ys = [(torch.randn(N, 1, H, W, C, device=dev, generator=g) * 1.5 + 0.2).half() for _ in range(R)]
    gamma, beta = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.3
    M = N * H * W
    yf = ys[0].float().view(M, C)
    stats = torch.stack([yf.double().sum(0).float(), (yf.double() ** 2).sum(0).float()]).view(1, 2, C)
    co = ops.bn_finalize(stats, C, gamma, beta, None, None, None, count_host=M)
    pa = [ops.maxpool2d_fwd(y, want_arg=True, bn=co) for y in ys]
    dps = [torch.randn(pa[0][0].shape, device=dev, generator=g).half() for _ in range(R)]
    mb = lambda *ts: sum(t.numel() * t.element_size() for t in ts) / 1e6
    rows = [
        ("maxpool2d_fwd (bn + relu + pool)", lambda i: ops.maxpool2d_fwd(ys[i % R], want_arg=True, bn=co), mb(ys[0], pa[0][0], pa[0][1])),
        ("bn-backward sums from y + tap map", lambda i: ops._maxpool_bn_bwd_sums(ys[i % R], pa[i % R][1], dps[i % R], co, gamma, None),
         mb(ys[0], pa[0][1], dps[0])),
        ("bn-backward sums from pooled", lambda i: ops._maxpool_bn_bwd_sums(ys[i % R], pa[i % R][1], dps[i % R], co, gamma, pa[i % R][0]),
         mb(pa[0][0], dps[0])),
    ]
    for name, fn, mbytes in rows:
        us = timeit(fn, 2 * R)
        print(f"{name:40s} {us:7.1f} us   {mbytes:7.1f} MB algorithmic   {mbytes / us / 1e3 * 1e3:6.2f} GB/s x1e3 = {mbytes / us:5.2f} TB/s")


if __name__ == "__main__":
    main()
