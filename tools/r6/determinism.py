#!/usr/bin/env python
"""Where does a replayed bench step stop being bit-reproducible?  One net, graphs captured once; the state (parameters, buffers, Adam
moments, step record) is restored and the same replays run again: per step, compare the loss and a checksum of every parameter."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa
from tricolo_amd import ops, parallel
from tricolo_amd.data import synthetic as syn

a = bench.parse_args([])
dev = torch.device("cuda:0")
net, cfg = bench.build_net(a, "f16", dev)
opt = net.configure_optimizers()
opt.prepare(captures=8)
NG = int(os.environ.get("NG", "1"))
batches = bench.make_batches(a, 0, dev, NG)
ops.DEBUG_KEEP = {}
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for b in batches[:2]:
        parallel.dp_training_step(net, b, opt)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
graphs, outs, pool = [], [], None
for b in batches:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=pool):
        outs.append(parallel.dp_training_step(net, b, opt)["train_loss/total_loss"])
    pool = pool or g.pool(); graphs.append(g)
torch.cuda.synchronize()
snap = (opt._flat_p.clone(), opt._flat_m.clone(), opt._flat_v.clone(), opt._step_dev.clone(), [b.clone() for b in net.buffers()])
names = [n for n, _ in net.named_parameters()]
def restore():
    with torch.no_grad():
        opt._flat_p.copy_(snap[0]); opt._flat_m.copy_(snap[1]); opt._flat_v.copy_(snap[2]); opt._step_dev.copy_(snap[3])
        for b, q in zip(net.buffers(), snap[4]): b.copy_(q)
    torch.cuda.synchronize()
def run(n):
    rec = []
    for i in range(n):
        graphs[i % NG].replay()
        torch.cuda.synchronize()
        rec.append((outs[i % NG].item(), [p.detach().clone() for p in net.parameters()], {k: v.detach().clone() for k, v in ops.DEBUG_KEEP.items()}))
    return rec
N = int(os.environ.get("N", "6"))
restore(); ref = run(N)
bad = 0
for rep in range(int(os.environ.get("REPS", "6"))):
    restore(); cur = run(N)
    for i, (r, c) in enumerate(zip(ref, cur)):
        pd = [(names[k], int((r[1][k] != c[1][k]).sum().item()), float((r[1][k] - c[1][k]).abs().max().item())) for k in range(len(names))
              if not torch.equal(r[1][k], c[1][k])]
        if r[0] != c[0] or pd:
            dbg = [(k, int((r[2][k] != c[2][k]).sum().item())) for k in r[2] if not torch.equal(r[2][k], c[2][k])]
            if "gru_dgh" in r[2] and not torch.equal(r[2]["gru_dgh"], c[2]["gru_dgh"]):
                B_ = a.per_gpu_batch
                d = (r[2]["gru_dgh"] != c[2]["gru_dgh"]).view(2, -1, B_, 3, 128)          # [dir][t][b][gate][unit]
                for dr in range(2):
                    idx = d[dr].nonzero()
                    if idx.numel() == 0: continue
                    ts = idx[:, 0]
                    t_org = int(ts.max().item()) if dr == 0 else int(ts.min().item())       # backward-in-time origin: last t (dir 0) / first t (reverse)
                    at = idx[ts == t_org]
                    rows = sorted(set(at[:, 1].tolist())); units = sorted(set(at[:, 3].tolist())); gates_ = sorted(set(at[:, 2].tolist()))
                    vals = (r[2]["gru_dgh"].view(2, -1, B_, 3, 128)[dr, t_org] - c[2]["gru_dgh"].view(2, -1, B_, 3, 128)[dr, t_org]).abs().max().item()
                    print(f"   dir {dr}: origin t={t_org}, rows {rows}, gates {gates_}, {len(units)} units {units[:20]}, max |diff| {vals:.3e}, steps affected {len(set(ts.tolist()))}")
            print(f"rep {rep} step {i}: loss {r[0]!r} vs {c[0]!r}; {len(pd)} params differ: {[x[0] for x in pd[:8]]}; intermediates that differ: {dbg}")
            bad += 1
            break
print("non-reproducible reps:", bad)
