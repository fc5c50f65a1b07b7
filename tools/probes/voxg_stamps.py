#!/usr/bin/env python
"""In-kernel stamps of conv_voxg_kernel.  Needs a PROBE build of conv_voxg.o (the production build carries no stamp code):
    cd tricolo_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-inline-asm -DVOXG_PROBE -c conv_voxg.hip -o conv_voxg.o && \
    hipcc -shared -fPIC --offload-arch=gfx950 *.o -o ../libtricolo_hip.so        (then `make -B conv_voxg.o && make` restores production)
Per workgroup (wave 0): shader-clock stamps at entry (0), after the mask / ranking / zero fill (1), before the chunk loop (2), after it (3),
after the epilogue stores (4), at exit (5); active rows (6); 100 MHz wall clock at stamp 1 (7)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tricolo_amd import ops
from tricolo_amd.data import synthetic as syn
import torch.nn.functional as F

dev = torch.device("cuda:0")
print("ablation bits:", os.environ.get("TRICOLO_VOXG_ABL", "0"))
for V, B in ((32, 32), (64, 64)):
    batch = syn.make_batch(B, voxel_size=V, num_views=None, seed=syn.BASE_SEED + 2)
    locs = batch["voxels"]["locs"].long()
    m = torch.zeros(B, 1, V, V, V)
    m[locs[:, 0], 0, locs[:, 1], locs[:, 2], locs[:, 3]] = 1
    chans = [3, 32, 64, 128, 256, 512]
    for l in range(5):
        D = V >> l
        cin, cout = chans[l], chans[l + 1]
        if D <= 8:
            g = ops.ConvGeom(B, (D, D, D), cin, cin, cout, (3, 3, 3), 1, (1, 1, 1), (27 * cin, cin, 1))
            fam = g.kernel_family[(False, 2)]
            assert (fam & 255) == 13
            M = B * D ** 3
            mask = torch.zeros((M + 31) // 32 * 32, dtype=torch.uint8)
            mask[:M] = m.reshape(-1).to(torch.uint8)
            mask = mask.to(dev)
            x = (torch.randn(B, D, D, D, cin) * m[:, 0, ..., None]).to(dev).half()
            w = torch.randn(cout, 3, 3, 3, cin, device=dev) * 0.05
            packed = ops.pack_weight(w, g, "f16")
            nwg = 4096
            dbg = torch.zeros((nwg, 16), dtype=torch.int64, device=dev)
            os.environ["TRICOLO_VOXG_DBG"] = str(dbg.data_ptr())
            for _ in range(3):
                dbg.zero_()
                y, st = ops.conv_fwd(x, g, packed, row_mask=mask, want_stats=True)
            torch.cuda.synchronize()
            d = dbg.cpu().numpy()
            d = d[d[:, 0] != 0]
            rows = d[:, 6]
            ne = rows > 0
            seg = [d[ne, i + 1] - d[ne, i] for i in range(5)]
            print(f"{V}^3 B{B} L{l} {D}^3 {cin}->{cout}: ct {(fam >> 8) & 255} spu {(fam >> 24) & 127}; {len(d)} workgroups, {ne.sum()} non-empty, rows/wg median {np.median(rows[ne]):.0f} max {rows.max()}")
            names = ["setup (mask, rank, zero fill)", "first slab commit", "chunk loop", "epilogue", "statistics"]
            for nme, sg in zip(names, seg):
                print(f"      {nme:32s} median {np.median(sg):8.0f}  p90 {np.percentile(sg, 90):8.0f}  max {sg.max():8.0f} cycles")
            fine = [("entry -> mask requested", 0, 8), ("ring loads issued", 8, 9), ("zero fill issued", 9, 10), ("mask arrived (ballot)", 10, 11),
                    ("rank barrier", 11, 12), ("tables + barrier", 12, 1)]
            for nme, i0, i1 in fine:
                sg = d[ne, i1] - d[ne, i0]
                print(f"        . {nme:30s} median {np.median(sg):8.0f}  p90 {np.percentile(sg, 90):8.0f} cycles")
            tot = d[ne, 5] - d[ne, 0]
            print(f"      total                            median {np.median(tot):8.0f}  p90 {np.percentile(tot, 90):8.0f}  max {tot.max():8.0f} cycles;  start spread {(d[:, 7].max() - d[:, 7].min()) / 100.0:.2f} us")
        m = F.max_pool3d(m, 2)
