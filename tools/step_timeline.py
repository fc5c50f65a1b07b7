#!/usr/bin/env python
"""Where the wall time of one training step goes, measured INSIDE the replayed HIP graph: device-side timestamps
(tri_debug_stamp, the 100 MHz wall clock) at the tower / stage boundaries of the step, no profiler attached.

    python tools/step_timeline.py [bench.py workload flags, e.g. --config 3 --per-gpu-batch 32 --precision f16]

Prints every stamp of the median replay, sorted by time, relative to `step.start`."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    a = bench.parse_args() if hasattr(bench, "parse_args") else None
    if a is None:
        raise SystemExit("bench.parse_args() not found")
    from tricolo_amd import ops, parallel
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    net, cfg = bench.build_net(a, a.precision, device)
    opt = net.configure_optimizers()
    opt.prepare()
    batch = bench.make_batches(a, 0, device, 1)[0]

    def step():
        return parallel.dp_training_step(net, batch, opt)["train_loss/total_loss"]

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    buf = torch.zeros((256,), dtype=torch.int64, device=device)
    ops.TIMELINE = {"buf": buf, "names": [], "fine": os.environ.get("TRICOLO_FINE_STAMPS", "0") == "1"}   # fine: a stamp behind every unit of the image tower
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
        ops.stamp("step.end")
    names = list(ops.TIMELINE["names"])
    ops.TIMELINE = None
    runs = []
    for i in range(60):
        g.replay()
        if i >= 40:
            torch.cuda.synchronize()
            runs.append(buf[:len(names)].cpu().clone())
    torch.cuda.synchronize()
    i0, i1 = names.index("step.start"), names.index("step.end")
    spans = sorted((int(r[i1] - r[i0]), k) for k, r in enumerate(runs))
    r = runs[spans[len(spans) // 2][1]]
    t0 = int(r[i0])
    rows = sorted(((int(r[k]) - t0) / 100.0, n) for k, n in enumerate(names))      # 100 MHz -> us
    print(f"# {a.precision}, per-GPU batch {a.per_gpu_batch}, median of {len(runs)} replays: step.start -> step.end "
          f"{(int(r[i1]) - t0) / 100.0:.1f} us (includes ~{len(names)} stamp launches)")
    prev = {}
    for t, n in rows:
        tower = n.split(".")[0]
        d = t - prev.get(tower, 0.0)
        prev[tower] = t
        print(f"{t:9.1f} us  (+{d:8.1f} in {tower:6s})  {n}")


if __name__ == "__main__":
    main()
