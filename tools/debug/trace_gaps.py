"""Summarise a rocprofv3 kernel trace of bench.py: per-step wall time, busy time and the largest idle gaps of the last timed steps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# find adam kernels as step delimiters
idx = [i for i, e in enumerate(ev) if e[2].startswith("adam_kernel") or e[2].startswith("adam_seg")]
print("kernels", len(ev), "adam launches", len(idx))
last = idx[-12:-2]
for a, b in zip(last[:-1], last[1:]):
    seg = ev[a + 1:b + 1]
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, _ in seg[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps = []
    end = seg[0][1]
    for s, e, n in seg[1:]:
        if s > end: gaps.append((s - end, n[:50]))
        end = max(end, e)
    gaps.sort(reverse=True)
    print(f"step wall {(t1 - t0) / 1e3:8.1f} us  busy {busy / 1e3:8.1f} us  launches {len(seg)}  top gaps {[(round(g / 1e3, 1), n) for g, n in gaps[:5]]}")
names = collections.Counter(e[2][:60] for e in ev if "ccl" in e[2].lower() or "nccl" in e[2].lower())
print("rccl kernels", names.most_common(5))
