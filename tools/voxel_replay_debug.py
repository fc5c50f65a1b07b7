"""Debug aid: per-tensor error of the voxel tower's parameter gradients against the forced-routing float64 replay (tests/replay.py),
for a precision mode and both row-selection paths (TRICOLO_VOXEL_COMPACT = 1 / 0), plus optional A/B switches from the environment.
    python tools/voxel_replay_debug.py f16"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests.replay import voxel_forced_replay  # noqa: E402
from tricolo_amd import ops  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
ops.set_default_precision(prec)
res = {}
for compact in ("1", "0"):
    os.environ["TRICOLO_VOXEL_COMPACT"] = compact
    table, unmatched, zdiff, g, rg = voxel_forced_replay(32, 8, want_grads=True)
    res[compact] = (table, g)
    print(f"--- {prec} compact={compact}: |dz| {zdiff:.2e}, unmatched windows {unmatched}")
    for n, v in table.items():
        cos = float((g[n] * rg[n]).sum() / (g[n].norm() * rg[n].norm()))
        print(f"  {n:28s} rel L2 {v:.3e}   norm ratio {float(g[n].norm() / rg[n].norm()):.5f}   1-cos {1 - cos:.2e}")
print("--- compact vs masked")
for n in res["1"][0]:
    a, b = res["1"][1][n], res["0"][1][n]
    print(f"  {n:28s} {float((a - b).norm() / a.norm()):.3e}")
