#!/usr/bin/env python
"""Unprofiled per-entry-point GPU time of one training step: every C-ABI call of eager, stream-serialised steps bracketed by HIP events
(tricolo_amd._C.install_timing_proxy), the event-pair cost of an empty launch subtracted.  Complements rocprofv3, which reads every
kernel shorter than ~4 us as ~4.5 us.      python tools/kernel_times.py [bench.py workload flags]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    a = bench.parse_args()
    from tricolo_amd import _C, ops, parallel
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    net, cfg = bench.build_net(a, a.precision, device)
    opt = net.configure_optimizers()
    opt.prepare()
    batch = bench.make_batches(a, 0, device, 1)[0]
    net.overlap_towers = False
    for side_name in ("_side", "_side_ds", "_side_prep"):
        if net.image_encoder is not None and getattr(net.image_encoder, side_name, None) is not None:
            getattr(net.image_encoder, side_name).enabled = False

    def step():
        return parallel.dp_training_step(net, batch, opt)["train_loss/total_loss"]
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t = ops.KernelTimer()
    torch.cuda._sleep(int(20e6))
    ovh = t.calibrate()
    proxy = _C.install_timing_proxy()
    nrep = 3
    for _ in range(nrep):
        torch.cuda._sleep(int(80e6))
        step()
    torch.cuda.synchronize()
    if os.environ.get("KT_TRACE") == "1":                           # the calls of the LAST step in issue order (eager, one stream)
        per = len(proxy.records) // nrep
        acc = 0.0
        for name, e0, e1 in proxy.records[-per:]:
            d = max(e0.elapsed_time(e1) - ovh, 0.0)
            acc += d
            print(f"{acc * 1e3:9.1f} us  {d * 1e3:7.1f}  {name}")
    agg = {}
    for name, e0, e1 in proxy.records:
        d = agg.setdefault(name, [0, 0.0])
        d[0] += 1
        d[1] += max(e0.elapsed_time(e1) - ovh, 0.0)
    tot = sum(v[1] for v in agg.values()) / nrep
    print(f"# {a.precision}, config {a.config}, per-GPU batch {a.per_gpu_batch}: {tot * 1e3:.0f} us of entry-point time per step (event overhead {ovh * 1e3:.1f} us subtracted per call)")
    for name, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:36s} {n // nrep:4d} calls  {ms / nrep * 1e3:8.1f} us/step  {ms / n * 1e3:7.1f} us avg")


if __name__ == "__main__":
    main()
