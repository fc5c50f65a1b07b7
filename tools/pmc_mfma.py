#!/usr/bin/env python
"""MFMA utilisation per kernel from a rocprofv3 PMC pass (north star: "evidenced by rocprof ... MFMA utilisation against gfx950 peak").

    cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python3 bench.py --steps 3 --warmup 2 \
        --modes "" --no-cpu-baseline
    python tools/pmc_mfma.py <dir> > profiles/r4/pmc_mfma_f16.json

SQ_VALU_MFMA_BUSY_CYCLES counts, summed over all SIMDs, the cycles the matrix pipe was busy (16 per v_mfma_f32_16x16x32_*);
GRBM_GUI_ACTIVE the busy cycles of the dispatch summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS section).  Utilisation of a kernel =
MFMA busy cycles / (GUI_ACTIVE / 8 x 1,024 SIMDs); mfma_tflops_at_2p4GHz = utilisation x 2.5 PF (what the pipe delivered per clock,
independent of the clock the chip held).  Kernel names as tools/pmc_traffic.py merges them."""
import collections
import csv
import glob
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import demangle, symbol  # noqa: E402


def main():
    d = sys.argv[1]
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    names = demangle(sorted({r["Kernel_Name"] for r in rows}))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in rows:
        k = symbol(names[r["Kernel_Name"]])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
    out = {}
    for k, v in per.items():
        gui, mfma = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if gui <= 0 or mfma <= 0:
            continue
        util = mfma / (gui / 8.0 * 1024.0)
        out[k] = {"dispatches": cnt[k], "mfma_busy_cycles_per_launch": int(mfma / cnt[k]), "gui_active_per_launch": int(gui / cnt[k]),
                  "mfma_utilisation": round(util, 4), "mfma_tflops_at_2p4GHz": round(util * 2500.0, 1)}
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["dispatches"]))
    json.dump({"method": __doc__.split("\n\n")[2], "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
