#!/usr/bin/env python
"""Does a replayed HIP graph accept a fork taken from a stream that is itself a forked branch (main -> B -> C, C joined into B, B into main)?"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def body2(b, c, x, y, z):
    """C forked from B but joined into MAIN"""
    main = torch.cuda.current_stream()
    b.wait_stream(main)
    x.add_(1)
    with torch.cuda.stream(b):
        y.add_(1)
        c.wait_stream(b)
        with torch.cuda.stream(c):
            z.add_(1)
        y.mul_(2)
    main.wait_stream(b)
    main.wait_stream(c)
    x.add_(y)
    x.add_(z)


def body(b, c, x, y, z):
    main = torch.cuda.current_stream()
    b.wait_stream(main)
    x.add_(1)
    with torch.cuda.stream(b):
        y.add_(1)
        c.wait_stream(b)
        with torch.cuda.stream(c):
            z.add_(1)
        y.mul_(2)
        b.wait_stream(c)
        y.add_(z)
    main.wait_stream(b)
    x.add_(y)


if __name__ == "__main__":
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    b, c, cap = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    x, y, z = (torch.zeros(1024, device=dev) for _ in range(3))
    if len(sys.argv) > 1 and sys.argv[1] == "2":
        body = body2
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        body(b, c, x, y, z)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        body(b, c, x, y, z)
    print("captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replayed", float(x[0]), float(y[0]), float(z[0]))
