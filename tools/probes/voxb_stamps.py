#!/usr/bin/env python
"""Per-phase cycle sums of conv_voxb_kernel (PROBE build of conv_voxg.o, see voxg_stamps.py)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tricolo_amd import ops
from tricolo_amd.data import synthetic as syn
import torch.nn.functional as F
dev = torch.device("cuda:0")
for V, B in ((32, 32), (64, 64)):
    batch = syn.make_batch(B, voxel_size=V, num_views=None, seed=syn.BASE_SEED + 2)
    locs = batch["voxels"]["locs"].long()
    m = torch.zeros(B, 1, V, V, V)
    m[locs[:, 0], 0, locs[:, 1], locs[:, 2], locs[:, 3]] = 1
    m = F.max_pool3d(m, 2)
    D = V // 2
    g = ops.ConvGeom(B, (D, D, D), 32, 32, 64, (3, 3, 3), 1, (1, 1, 1), (27 * 32, 32, 1))
    assert (g.kernel_family[(False, 2)] & 255) == 14
    M = B * D ** 3
    mask = m.reshape(-1).to(torch.uint8).to(dev)
    x = (torch.randn(B, D, D, D, 32) * m[:, 0, ..., None]).to(dev).half()
    w = torch.randn(64, 3, 3, 3, 32, device=dev) * 0.05
    packed = ops.pack_weight(w, g, "f16")
    dbg = torch.zeros((1024, 16), dtype=torch.int64, device=dev)
    os.environ["TRICOLO_VOXG_DBG"] = str(dbg.data_ptr())
    for _ in range(3):
        dbg.zero_()
        y, st = ops.conv_fwd(x, g, packed, row_mask=mask, want_stats=True)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    d = d[d[:, 8] != 0]
    names = ["wait S1", "clear + ballots", "rank barrier", "tables + S2", "slab load + S3", "MFMA passes + stores", "-", "loop top"]
    print(f"{V}^3 B{B} level 1 ({D}^3): {len(d)} workgroups, bricks/wg {np.median(d[:, 11]):.0f}, non-empty/wg {np.median(d[:, 9]):.0f}, rows/wg {np.median(d[:, 10]):.0f}; total cycles median {np.median(d[:, 8]):.0f} max {d[:, 8].max()}")
    for i, n in enumerate(names):
        if n != "-":
            print(f"      {n:24s} median {np.median(d[:, i]):9.0f}   per brick {np.median(d[:, i] / np.maximum(d[:, 11], 1)):7.0f}")
