#!/usr/bin/env python
"""Which HIP streams do the towers' side streams really are?  torch hands torch.cuda.Stream() objects out of a pool of 32 per device and
priority, round-robin: two SideStream objects created far apart can be the SAME stream - and then serialise in a captured graph."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    a = bench.parse_args()
    from tricolo_amd import parallel
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    net, cfg = bench.build_net(a, a.precision, device)
    opt = net.configure_optimizers()
    opt.prepare()
    batch = bench.make_batches(a, 0, device, 1)[0]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            parallel.dp_training_step(net, batch, opt)
    torch.cuda.synchronize()
    names = {"capture/warm-up stream": s, "tower text": net._side_streams[0], "tower voxel": net._side_streams[1]}
    enc = net.image_encoder
    for n in ("_side", "_side_ds", "_side_prep"):
        st = getattr(enc, n).stream
        if st is not None:
            names["image " + n] = st
    for k, v in names.items():
        print(f"{k:28s} stream id {v.stream_id:4d}  handle {v.cuda_stream:#x}")
    more = [torch.cuda.Stream() for _ in range(40)]
    print("next 40 streams from the pool:", sorted({m.cuda_stream for m in more}).__len__(), "distinct handles")


if __name__ == "__main__":
    main()
