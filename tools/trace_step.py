#!/usr/bin/env python
"""One training step out of a rocprofv3 --kernel-trace CSV: every kernel of the LAST complete replayed step in start order with its
start offset, duration, queue and the idle gap of its queue in front of it (the critical chain of the step reads off this list).
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --modes "" --no-cpu-baseline --steps 10 --warmup 3 --windows 1
    python tools/trace_step.py <dir with *_kernel_trace.csv> [--step -2 | --shortest]   (--shortest: a replayed step, not an eager warm-up one)"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    m = re.match(r"_Z\d+([a-z0-9_]+?)I", name)
    return (m.group(1) if m else name)[:44]


def main():
    d = sys.argv[1]
    which = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -2
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    cols = None
    for f in files:
        for r in csv.DictReader(open(f)):
            cols = cols or list(r.keys())
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if re.search(r"\badam_(seg_)?kernel", r[2])]
    if len(ends) < 2:
        raise SystemExit(f"{len(rows)} kernel rows, {len(ends)} Adam launches: no complete step in the trace")
    print(f"# {len(rows)} kernel rows, {len(ends)} Adam launches; columns {cols}")
    if "--shortest" in sys.argv:                         # the shortest complete step (a replayed one, not an eager warm-up step)
        which = min(range(1, len(ends)), key=lambda i: rows[ends[i]][1] - rows[ends[i - 1] + 1][0])
    lo, hi = ends[which - 1] + 1, ends[which] + 1
    step = rows[lo:hi]
    t0 = step[0][0]
    last_end = {}
    print(f"# step of {len(step)} kernels, {(step[-1][1] - t0) / 1000:.1f} us from first start to last end")
    busy = 0
    for s, e, n, q in step:
        gap = (s - last_end[q]) / 1000 if q in last_end else 0.0
        last_end[q] = e
        busy += e - s
        print(f"{(s - t0) / 1000:9.1f} {(e - s) / 1000:7.1f} us  q{q:>3s} gap {gap:6.1f}  {short(n)}")
    print(f"# summed kernel time {busy / 1000:.1f} us")


if __name__ == "__main__":
    main()
