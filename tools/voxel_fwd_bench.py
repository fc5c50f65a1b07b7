#!/usr/bin/env python
"""3D-conv forward of the voxel tower at the sizes that matter: the five SubMConv3d forwards timed by HIP
events, dense / executed / active-row FLOPs and % of the 2.5 PF dense MFMA peak, at 32^3 x B64 (BASELINE config 2) and
64^3 x B64 (config 5), per precision mode.

    python tools/voxel_fwd_bench.py [--out profiles/r2/voxel_fwd.txt]
"""
import argparse
import json
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tricolo_amd.data import synthetic as syn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--json", default="")
    ap.add_argument("--modes", default="f16,bf16")
    ap.add_argument("--shapes", default="32x32,32x64,64x64", help="comma list of <voxel grid>x<batch>")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows, docs = [], {}
    for V, B in [tuple(int(v) for v in sh.split("x")) for sh in a.shapes.split(",")]:
        batch = syn.batch_to_device(syn.make_batch(B, voxel_size=V, num_views=None, seed=syn.BASE_SEED + 2), dev)
        for mode in a.modes.split(","):
            args = types.SimpleNamespace(text="BiGRUEncoder", image=None, voxel="SparseCNNEncoder", voxel_size=V, num_views=6, image_size=128)
            net, _ = bench.build_net(args, mode, dev)
            bench.voxel_fwd_roofline(net, batch, B, nrep=2)          # warm-up (lazy buffers, first-touch)
            r = bench.voxel_fwd_roofline(net, batch, B, nrep=5)
            docs[f"{V}^3 B{B} {mode}"] = r
            rows.append(f"{V}^3 x B{B:<3d} {mode:5s} total {r['ms']:.4f} ms  dense {r['achieved_dense_equivalent']:7.1f} TF ({100 * r['frac_dense_equivalent']:.2f} %)  "
                        f"executed {r['achieved']:7.1f} TF ({100 * r['frac']:.2f} % of 2.5 PF)  active-row {r['active_row_flops'] / r['ms'] / 1e9:7.1f} TF  "
                        f"L0 HBM {r['level0_hbm']['achieved']:.0f} GB/s  | raw events {r['ms_raw']:.4f} ms ({100 * r['frac']:.2f} %), "
                        f"back to back {r['ms_back_to_back']:.4f} ms ({100 * r['frac_back_to_back']:.2f} %)")
            for l in r["levels"]:
                rows.append(f"    L{l['level']} {l['kernel']:34s} {l['grid']:3d}^3 {l['cin']:3d}->{l['cout']:3d}  active {l['active_sites']:8d}/{l['sites']:9d} sites, "
                            f"{l['executed_tiles']:6d}/{l['tiles']:6d} tiles{'*' if l['compact_rows'] else ' '}  {l['ms']:.4f} ms  dense {l['dense_tflops']:7.1f}  executed {l['executed_tflops']:7.1f}  "
                            f"active-row {l['active_row_tflops']:6.1f} TF  {l['algorithmic_hbm_gbs']:7.1f} GB/s  b2b {l['ms_back_to_back']:.4f} ms")
            del net
            torch.cuda.empty_cache()
    text = "\n".join(rows)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            f.write("# tools/voxel_fwd_bench.py: five SubMConv3d forwards (sparse_cnn.py:12-32), HIP-event timed, median of 5\n" + text + "\n")
    if a.json:
        with open(a.json, "w") as f:
            json.dump(docs, f, indent=1)


if __name__ == "__main__":
    main()
