#!/usr/bin/env python
"""Per-layer timing of the conv kernels (fwd / dgrad / wgrad) over every layer geometry of the bench workload.
Tuning harness: python tools/conv_layers_bench.py [--precision bf16] [--batch 32] [--only resnet|voxel]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tricolo_amd import ops  # noqa: E402
from tricolo_amd.data import synthetic as syn  # noqa: E402


COLD = None                                                 # --cold: a 640 MB buffer rewritten in front of every timed call


def time_it(fn, iters=10):
    """GPU time per call: the calls are captured into a HIP graph first, so host launch overhead (tens of us per
    Python-level op) does not leak into the measurement of short kernels.  --cold: every call runs behind a pass over a buffer
    larger than L2 + Infinity Cache (as in the training step, where a layer's weights were packed megabytes of traffic ago);
    the flush passes alone are timed the same way and subtracted."""
    if COLD is not None:
        def both():
            COLD.add_(1.0)
            fn()
        return _time_graph(both, iters) - _time_graph(lambda: COLD.add_(1.0), iters)
    return _time_graph(fn, iters)


def _time_graph(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    ts = []
    for _ in range(5 if COLD is not None else 3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters)
    return sorted(ts)[len(ts) // 2] if COLD is not None else min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--views", type=int, default=6)
    ap.add_argument("--image", type=int, default=128)
    ap.add_argument("--voxel", type=int, default=32)
    ap.add_argument("--only", default="")
    ap.add_argument("--layers", default="", help="regex over the layer names")
    ap.add_argument("--cold", action="store_true", help="flush L2 / Infinity Cache in front of every timed call")
    ap.add_argument("--no-wgrad", action="store_true")
    args = ap.parse_args()
    if args.cold:
        global COLD
        COLD = torch.zeros(160 * 1024 * 1024, device="cuda")
    dev = "cuda"
    prec = args.precision
    rows = []
    layers = []
    N = args.batch * args.views
    S = args.image
    if args.only in ("", "resnet"):
        layers.append(("stem7x7", ops.ConvGeom(N, (1, S, S), 3, 4, 64, (1, 7, 7), 2, (0, 3, 3), (147, 1, 49)), None))
        h = S // 4
        inpl = 64
        for planes, stride in ((64, 1), (128, 2), (256, 2), (512, 2)):
            layers.append((f"l{planes}.c3x3s{stride}", ops.ConvGeom(N, (1, h, h), inpl, inpl, planes, (1, 3, 3), stride, (0, 1, 1), (inpl * 9, 1, 9)), None))
            if stride != 1:
                layers.append((f"l{planes}.ds1x1", ops.ConvGeom(N, (1, h, h), inpl, inpl, planes, (1, 1, 1), stride, (0, 0, 0), (inpl, 1, 1)), None))
            h //= stride
            layers.append((f"l{planes}.c3x3s1 x3", ops.ConvGeom(N, (1, h, h), planes, planes, planes, (1, 3, 3), 1, (0, 1, 1), (planes * 9, 1, 9)), None))
            inpl = planes
    if args.only in ("", "voxel"):
        batch = syn.make_batch(args.batch, voxel_size=args.voxel, num_views=None, seed=5)
        locs = batch["voxels"]["locs"]
        V = args.voxel
        m = torch.zeros(args.batch, 1, V, V, V)
        m[locs[:, 0].long(), 0, locs[:, 1].long(), locs[:, 2].long(), locs[:, 3].long()] = 1
        chans = [3, 32, 64, 128, 256, 512]
        for l in range(5):
            D = V >> l
            cin, cout = chans[l], chans[l + 1]
            cs = 4 if cin == 3 else cin
            g = ops.ConvGeom(args.batch, (D, D, D), cin, cs, cout, (3, 3, 3), 1, (1, 1, 1), (27 * cin, cin, 1))
            M = args.batch * D ** 3
            mk = torch.zeros((M + 31) // 32 * 32, dtype=torch.uint8)
            mk[:M] = m.reshape(-1).to(torch.uint8)
            layers.append((f"vox_l{l} occ={m.mean().item():.2f}", g, mk.to(dev)))
            m = torch.nn.functional.max_pool3d(m, 2)
    if args.only in ("", "linear"):
        layers.append(("gru_xproj", ops.ConvGeom(96 * args.batch, (1, 1, 1), 256, 256, 768, (1, 1, 1), 1, (0, 0, 0), (256, 1, 1)), None))
        layers.append(("mlp512", ops.ConvGeom(args.batch, (1, 1, 1), 512, 512, 512, (1, 1, 1), 1, (0, 0, 0), (512, 1, 1)), None))
    # last column pair: the layer's weight gradient as the training step runs it - FOUR copies of the layer queued in a WgradBatch (one
    # grouped partial launch per tile family + one grouped reduce), time per layer
    print(f"{'layer':28s} {'M':>8s} {'K':>6s} {'N':>4s} | {'fwd ms':>8s} {'TF':>7s} | {'dgrad ms':>8s} {'TF':>7s} | {'wgrad ms':>8s} {'TF':>7s} | {'x4 grouped':>10s} {'TF':>7s}")
    tot = [0.0, 0.0, 0.0]
    import re
    for name, g, mask in layers:
        if args.layers and not re.search(args.layers, name):
            continue
        ID, IH, IW = g.in_grid
        OD, OH, OW = g.out_grid
        x = torch.randn(g.B, ID, IH, IW, g.cin_stored, device=dev)
        dy = torch.randn(g.B, OD, OH, OW, g.cout, device=dev)
        if mask is not None:
            x = x * mask[:g.M_in].view(g.B, ID, IH, IW, 1).float()
            dy = dy * mask[:g.M].view(g.B, OD, OH, OW, 1).float()
        x, dy = x.to(ops.act_dtype(prec)), dy.to(ops.act_dtype(prec))     # bf16 mode stores activations as bf16
        w = torch.randn(g.cout * g.ntaps * g.cin, device=dev) * 0.05
        packed = ops.pack_weight(w, g, prec)
        t_f = time_it(lambda: ops.conv_fwd(x, g, packed, row_mask=mask, want_stats=True))
        t_d = float("nan")
        if g.cin == g.cin_stored and g.cin % 32 == 0:
            packed_t = ops.pack_weight(w, g, prec, transposed=True)
            t_d = time_it(lambda: ops.conv_dgrad(dy, g, packed_t, row_mask=mask))
        t_w = float("nan") if args.no_wgrad else time_it(lambda: ops.conv_wgrad(x, dy, g, w, prec, row_mask=mask))
        t_g = float("nan")
        if not args.no_wgrad and mask is None and x.dtype != torch.float32 and g.wgrad_group(ops._abf(x))[0]:
            def grouped():
                batch = ops.WgradBatch(dev)
                for _ in range(4):
                    ops.conv_wgrad(x, dy, g, w, prec, batch=batch)
                batch.flush()
            t_g = time_it(grouped) / 4
        fl = g.flops / 1e9
        mult = 3 if "x3" in name else 1
        tot[0] += t_f * mult
        tot[1] += (0 if t_d != t_d else t_d) * mult
        tot[2] += (0 if t_w != t_w else t_w) * mult
        print(f"{name:28s} {g.M:8d} {g.kpad:6d} {g.cout:4d} | {t_f:8.3f} {fl / t_f:7.1f} | {t_d:8.3f} {fl / t_d:7.1f} | {t_w:8.3f} {fl / t_w:7.1f} | {t_g:10.3f} {fl / t_g:7.1f}")
    print(f"totals (x3 layers weighted): fwd {tot[0]:.3f} ms  dgrad {tot[1]:.3f} ms  wgrad {tot[2]:.3f} ms  sum {sum(tot):.3f} ms")


if __name__ == "__main__":
    main()
