#!/usr/bin/env python
"""In-kernel stamps of conv_vox0_kernel.  Needs a PROBE build of the library (the production build carries no stamp code):
    cd tricolo_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-inline-asm -DVOX_PROBE -c conv_vox.hip -o conv_vox.o && \
    hipcc -shared -fPIC --offload-arch=gfx950 *.o -o ../libtricolo_hip.so        (then `make -B conv_vox.o && make` restores production)
TRICOLO_VOX_ABL=<bits> in such a build ablates the run loop (1 no MFMA, 2 no stores, 4 no slab loads, 8 no fragment reads, 16 prologue only).
Output: per-workgroup shader-clock stamps of wave 0 at entry, after the
mask test, after the slab fill and after the run loop, the number of active runs, and 100 MHz wall-clock start / end."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tricolo_amd import ops
from tricolo_amd.data import synthetic as syn

for V, B in ((32, 32), (64, 64)):
    dev = torch.device("cuda:0")
    batch = syn.batch_to_device(syn.make_batch(B, voxel_size=V, num_views=None, seed=syn.BASE_SEED + 2), dev)
    g = ops.ConvGeom(B, (V, V, V), 3, 4, 32, (3, 3, 3), 1, (1, 1, 1), (81, 3, 1))
    w = torch.randn(32, 3, 3, 3, 3, device=dev)
    packed = ops.pack_weight(w, g, "f16")
    x, mask = ops.voxel_scatter(batch["voxels"]["locs"], batch["voxels"]["feats"], B, V, dtype=torch.float16)
    nwg = g.num_mtiles[2]
    dbg = torch.zeros((nwg, 8), dtype=torch.int64, device=dev)
    os.environ["TRICOLO_VOX_DBG"] = str(dbg.data_ptr())
    for _ in range(3):
        y, st = ops.conv_fwd(x, g, packed, row_mask=mask, want_stats=True)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    runs = d[:, 4]
    ne = runs > 0
    t_mask, t_slab, t_loop = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
    wall = (d[:, 6].max() - d[:, 5].min()) / 100.0
    print(f"{V}^3 B{B}: {nwg} workgroups, {ne.sum()} non-empty, runs/non-empty {runs[ne].mean():.1f}; kernel span {wall:.1f} us")
    print(f"   cycles to mask test: median {np.median(t_mask):.0f} (empty wgs {np.median(t_mask[~ne]) if (~ne).any() else 0:.0f})")
    print(f"   slab fill: median {np.median(t_slab[ne]):.0f}  p90 {np.percentile(t_slab[ne], 90):.0f}")
    print(f"   run loop:  median {np.median(t_loop[ne]):.0f}  p90 {np.percentile(t_loop[ne], 90):.0f};  per run of wave 0: {np.median(t_loop[ne] / np.maximum(runs[ne] / 4, 1)):.0f} cycles")
    dur = (d[:, 6] - d[:, 5]) / 100.0
    print(f"   workgroup lifetime us: non-empty median {np.median(dur[ne]):.2f} p90 {np.percentile(dur[ne], 90):.2f}; empty median {np.median(dur[~ne]) if (~ne).any() else 0:.2f}")
    start = (d[:, 5] - d[:, 5].min()) / 100.0
    print(f"   start times us: p10 {np.percentile(start, 10):.1f} p50 {np.percentile(start, 50):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f}")
