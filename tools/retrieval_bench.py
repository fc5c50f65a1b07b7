#!/usr/bin/env python
"""Device ranking (tri_retrieval_topk) vs the reference's numpy ranking on the held-out set size of SURVEY.md 8d
(2,560 queries x 512 shapes x 512 dims): python tools/retrieval_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tricolo_amd import ops  # noqa: E402

rng = np.random.default_rng(0)
Nq, Ns, D = 2560, 512, 512
text = rng.standard_normal((Nq, D)).astype(np.float32)
shape = rng.standard_normal((Ns, D)).astype(np.float32)
lab = rng.integers(0, Ns, Nq).astype(np.int32)
t, s, l = (torch.from_numpy(a).cuda() for a in (text, shape, lab))
for _ in range(3):
    idx, sim, hit = ops.retrieval_topk(t, s, l, 5)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    idx, sim, hit = ops.retrieval_topk(t, s, l, 5)
b.record()
torch.cuda.synchronize()
gpu_ms = a.elapsed_time(b) / 10
t0 = time.perf_counter()
sims = np.dot(text.astype(np.float64), shape.T)                 # the reference's host algorithm (eval_retrieval.py:70-82)
order = np.flip(np.argsort(sims, axis=1), 1)
ref_idx = order[:, :5]
cpu_ms = (time.perf_counter() - t0) * 1e3
assert np.array_equal(idx.cpu().numpy(), ref_idx)
assert np.array_equal(hit.cpu().numpy(), np.argmax(order == lab[:, None], axis=1))
print(f"tri_retrieval_topk {gpu_ms:.3f} ms   numpy dot + argsort {cpu_ms:.1f} ms   ({Nq} x {Ns} x {D}, top-5 + ranks identical)")
