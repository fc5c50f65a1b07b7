#!/usr/bin/env python
"""Is the text tower's forward + backward bit-reproducible while another stream keeps the GPU busy?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tricolo_amd import ops
from tricolo_amd.data import synthetic as syn
from tricolo_amd.model.module.text_encoder.bigru import BiGRUEncoder
from oracle.recipe import fill_module

ops.set_default_precision("f16")
dev = "cuda"
net = BiGRUEncoder(syn.DEFAULT_VOCAB, 512).to(dev)
fill_module(net)
batch = syn.batch_to_device(syn.make_batch(32, voxel_size=None, num_views=None, seed=syn.BASE_SEED + 4), dev)
tok = batch["tokens"]
dz = torch.randn(32, 512, device=dev) * 1e-2
hooks = {}
def run():
    for p in net.parameters(): p.grad = None
    z = net(tok, batch)
    z.backward(dz)
    return [z.detach().clone()] + [p.grad.detach().clone() for p in net.parameters()]
names = ["z"] + [n for n, _ in net.named_parameters()]
ref = run(); torch.cuda.synchronize()
bg = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev, dtype=torch.float16); b = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
big = torch.zeros(64 << 20, device=dev)
mode = os.environ.get("BG", "mm")
bad = collections = 0
import collections as C
cnt = C.Counter()
for rep in range(int(os.environ.get("REPS", "300"))):
    with torch.cuda.stream(bg):
        if mode == "mm":
            for _ in range(3): torch.mm(a, b)
        elif mode == "mem":
            for _ in range(3): big.add_(1.0)
    cur = run()
    torch.cuda.synchronize()
    d = [names[k] for k in range(len(names)) if not torch.equal(ref[k], cur[k])]
    if d:
        bad += 1
        for n in d: cnt[n] += 1
print("BG", mode, "non-reproducible reps:", bad, dict(cnt))
