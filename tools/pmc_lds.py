#!/usr/bin/env python
"""LDS bank conflicts and MFMA-busy cycles per kernel from a rocprofv3 --pmc run:
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d <dir> -- python3 tools/voxel_fwd_bench.py --modes f16
    python tools/pmc_lds.py <dir> [name filter, comma separated]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
filt = sys.argv[2].split(",") if len(sys.argv) > 2 else ["vox", "conv_dma", "igemm"]
f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:80] + " grid=" + r["Grid_Size"]
    if not any(x in k for x in filt):
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
        cnt[k] += 1
for k, v in sorted(agg.items()):
    n = max(cnt[k], 1)
    a, c, m = v.get("SQ_LDS_IDX_ACTIVE", 0.0), v.get("SQ_LDS_BANK_CONFLICT", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    print("%-100s n=%3d  lds_active/launch=%12.0f  conflict_frac=%.3f  mfma_busy/launch=%12.0f" % (k, cnt[k], a / n, c / max(a, 1.0), m / n))
