"""Reference numbers for the held-out retrieval test (SURVEY.md 8d "Held-out retrieval set").  TEST INFRASTRUCTURE.

Trains the CPU oracle (restatement of the reference step, pinned to the reference by tests/golden) on the factor-based
synthetic set of tricolo_amd/data/synthetic.py from the deterministic recipe weights, with a fixed data order, and records
at several checkpoints the metrics and top-5 index matrix of the reference's compute_metrics on 512 UNSEEN shapes x 5
captions = 2,560 queries.  tests/test_gpu_modules.py::test_heldout_retrieval_rr1 trains the HIP path the same way and
compares.  Run in the build container (minutes of CPU):

    python -m oracle.make_heldout_rr [--steps 900] [--checkpoints 300,600,900]

Output: tests/golden/heldout_rr.npz
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import modules as om  # noqa: E402
from oracle.recipe import fill_module  # noqa: E402
from oracle.retrieval import collate_shape_embedding, compute_metrics_ref  # noqa: E402
from tricolo_amd.data import synthetic as syn  # noqa: E402

V, NV, S, B = 32, 2, 64, 32
TRAIN_SHAPES, HELD_SHAPES, HELD_CAPTIONS = 1024, 512, 5
SEED_TRAIN, SEED_HELD, SEED_ORDER = syn.BASE_SEED + 301, syn.BASE_SEED + 302, 7


def datasets():
    train = syn.make_factor_retrieval_set(TRAIN_SHAPES, 2, V, NV, S, seed=SEED_TRAIN, distinct=False)
    held = syn.make_factor_retrieval_set(HELD_SHAPES, HELD_CAPTIONS, V, NV, S, seed=SEED_HELD, distinct=True)
    return train, held


def batch_indices(rng, nshape):
    """One caption of 32 distinct training shapes (a duplicate shape inside a batch would be a false negative)."""
    sh = rng.choice(nshape, size=B, replace=False)
    return [2 * int(s) + int(rng.integers(0, 2)) for s in sh]


def data_sha(items):
    h = hashlib.sha256()
    for it in items[::37]:
        h.update(it["tokens"].tobytes())
        h.update(it["grid"].tobytes())
        h.update(it["imgs"].tobytes())
    return h.hexdigest()


def embed(model, items, device=None, bs=64):
    out = {"text": [], "image": [], "voxel": []}
    with torch.no_grad():
        for i in range(0, len(items), bs):
            b = syn.collate_items(items[i:i + bs], voxel=True, views=True)
            if device is not None:
                b = syn.batch_to_device(b, device)
            e = model(b)
            for k in out:
                out[k].append(e[f"{k}_features"].float().cpu().numpy())
    return {k: np.concatenate(v) for k, v in out.items()}


def metrics(e, items):
    ids = [f"shape{it['shape']:05d}" for it in items]
    return compute_metrics_ref(ids, e["text"], collate_shape_embedding(e["text"], e["image"], e["voxel"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=900)
    ap.add_argument("--checkpoints", default="300,600,900")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    cps = [int(x) for x in a.checkpoints.split(",")]
    torch.set_num_threads(a.threads)
    train, held = datasets()
    ref = om.TriCoLoRef(om.BiGRURef(syn.DEFAULT_VOCAB, 512), om.MVCNNRef(512, 512, "resnet18", NV), om.SparseCNNRef(V, 32, 512, 512))
    fill_module(ref)
    opt = torch.optim.Adam(ref.parameters(), lr=3.5e-4, weight_decay=1e-6)        # config/config.yaml:50-53
    rng = np.random.default_rng(SEED_ORDER)
    out = {"steps": np.int64(a.steps), "checkpoints": np.array(cps), "train_sha": data_sha(train), "held_sha": data_sha(held)}
    losses, t0 = [], time.time()
    for step in range(1, a.steps + 1):
        batch = syn.collate_items([train[i] for i in batch_indices(rng, TRAIN_SHAPES)], voxel=True, views=True)
        opt.zero_grad(set_to_none=True)
        loss, _, _ = ref.training_step(batch)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        if step % 50 == 0:
            print(f"step {step} loss {losses[-1]:.4f} ({time.time() - t0:.0f} s)", flush=True)
        if step in cps:
            ref.eval()
            m = metrics(embed(ref, held), held)
            ref.train()
            out[f"cp{step}/recall_rate"] = m["recall_rate"]
            out[f"cp{step}/ndcg"] = m["ndcg"]
            out[f"cp{step}/mrr"] = np.float64(m["mrr"])
            out[f"cp{step}/indices"] = m["indices"].astype(np.int16)
            out[f"cp{step}/labels"] = m["labels"].astype(np.int16)
            print(f"checkpoint {step}: RR@1 {100 * m['recall_rate'][0]:.2f} RR@5 {100 * m['recall_rate'][4]:.2f}", flush=True)
    out["losses"] = np.array(losses, dtype=np.float32)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "heldout_rr.npz"), **out)


if __name__ == "__main__":
    main()
