#!/usr/bin/env python
"""How a replayed HIP graph starts a forked branch: two independent chains of NA / NB kernels of dA / dB microseconds each (spin kernels),
chain A captured first on the capture stream, chain B on a side stream forked BEFORE chain A is issued.  Device-side stamps at the
start / end of each chain tell whether B's start waits for A's node COUNT (host / packet enqueue order), for A's DURATION (both folded
onto one queue) or for nothing.       python tools/probes/graph_fork.py"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tricolo_amd import ops  # noqa: E402


def run(NA, dA, NB, dB, order="AB", clock_mhz=2400.0):
    dev = torch.device("cuda:0")
    buf = torch.zeros((64,), dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()
    cap = torch.cuda.Stream()

    def chain(n, d, tag):
        ops.stamp(tag + ".start")
        for i in range(n):
            torch.cuda._sleep(int(d * clock_mhz))
            if i == 0:
                ops.stamp(tag + ".k1")
        ops.stamp(tag + ".end")

    def body():
        main = torch.cuda.current_stream()
        ops.stamp("t0")
        side.wait_stream(main)
        for which in order:
            if which == "A":
                chain(NA, dA, "A")
            else:
                with torch.cuda.stream(side):
                    chain(NB, dB, "B")
        main.wait_stream(side)
        ops.stamp("end")
    ops.TIMELINE = {"buf": buf, "names": []}
    cap.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        body()                                   # warm (eager)
    torch.cuda.synchronize()
    ops.TIMELINE = {"buf": buf, "names": []}
    with torch.cuda.graph(g, stream=cap):
        body()
    names = list(ops.TIMELINE["names"])
    ops.TIMELINE = None
    runs = []
    for i in range(30):
        g.replay()
        if i >= 10:
            torch.cuda.synchronize()
            runs.append(buf[:len(names)].cpu().clone())
    i0, i1 = names.index("t0"), names.index("end")
    spans = sorted((int(r[i1] - r[i0]), k) for k, r in enumerate(runs))
    r = runs[spans[len(spans) // 2][1]]
    t = {n: (int(r[k]) - int(r[i0])) / 100.0 for k, n in enumerate(names)}
    print(f"order {order}  A {NA:3d} x {dA:5.1f} us  B {NB:3d} x {dB:5.1f} us | A.start {t['A.start']:7.1f} A.k1 {t['A.k1']:7.1f} A.end {t['A.end']:7.1f} | "
          f"B.start {t['B.start']:7.1f} B.k1 {t['B.k1']:7.1f} B.end {t['B.end']:7.1f} | end {t['end']:7.1f}")


if __name__ == "__main__":
    torch.cuda.set_device(0)
    if os.environ.get("FORK_SHARED") == "1":           # is node release a shared serial resource?  chain A alone, then beside an equal chain B
        run(200, 1.0, 1, 1.0, "AB")
        run(200, 1.0, 200, 1.0, "AB")
        run(200, 1.0, 200, 1.0, "BA")
        run(100, 5.0, 1, 1.0, "AB")
        run(100, 5.0, 200, 1.0, "AB")
        run(100, 5.0, 200, 1.0, "BA")
        sys.exit(0)
    if os.environ.get("FORK_QUICK") == "1":
        run(50, 5.0, 10, 5.0, "AB")
        run(200, 1.0, 10, 5.0, "AB")
        sys.exit(0)
    for NA, dA in ((10, 5.0), (50, 5.0), (100, 5.0), (50, 1.0), (50, 20.0), (200, 1.0)):
        run(NA, dA, 10, 5.0, "AB")
    run(50, 5.0, 10, 5.0, "BA")
    run(50, 5.0, 50, 5.0, "AB")
