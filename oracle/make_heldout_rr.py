"""Reference numbers for the held-out retrieval test (SURVEY.md 8d "Held-out retrieval set").  TEST INFRASTRUCTURE.

Trains the REAL reference model - /root/reference's TriCoLoNet (tricolo_net.py:46-71: forward, _calculate_losses,
configure_optimizers) under the import shims of oracle/make_golden.py - on the factor-based synthetic set of
tricolo_amd/data/synthetic.py from the deterministic recipe weights, with a fixed data order, and records at several checkpoints
what the REAL compute_metrics (eval_retrieval.py:249-278) returns on 512 UNSEEN shapes x 5 captions = 2,560 queries: RR@k, NDCG,
MRR and the top-5 index matrix.  The restatement (oracle/modules.py + oracle/retrieval.py) is trained beside it from the same
weights on the same batches; its per-step losses and checkpoint metrics are stored too (`restated/...`), so
tests/test_oracle_golden.py can assert that the restatement reproduces the reference on this path, and
tests/test_gpu_modules.py::test_heldout_retrieval_rr1 trains the HIP path the same way and compares with the REFERENCE's numbers.
Run in the build container (tens of minutes of CPU; --restated-only skips the reference and keeps the round-2 behaviour):

    python -m oracle.make_heldout_rr [--steps 600] [--checkpoints 200,400,600]

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


def real_metrics(e, items):
    """The reference's own compute_metrics on the tuples tricolo_net.py:125-158 (_collate_output) would hand it: shape embedding =
    image + voxel features (tricolo_net.py:134-138), one tuple per caption."""
    import tempfile
    from tricolo.evaluation import eval_retrieval as er
    shape = e["image"] + e["voxel"]
    emb = {"caption_embedding_tuples": [(None, "synthetic", f"shape{it['shape']:05d}", e["text"][i], shape[i]) for i, it in enumerate(items)]}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)                                   # compute_metrics writes nearest.jsonl into the CWD (eval_retrieval.py:292)
        try:
            pr = er.compute_metrics("Text2ShapeChairTable", emb)
            (tm, sm, labels, fit_labels, _, _, _) = er.construct_embeddings_matrix("Text2ShapeChairTable", emb)
            _, indices, _ = er.compute_nearest_neighbors(sm, tm, 5)
        finally:
            os.chdir(cwd)
    return {"recall_rate": np.asarray(pr["recall_rate"]), "ndcg": np.asarray(pr["ndcg"]), "mrr": float(pr["mrr"]), "indices": np.asarray(indices),
            "labels": np.asarray(labels)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--checkpoints", default="200,400,600")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--restated-only", action="store_true")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "heldout_rr.npz"))
    a = ap.parse_args()
    cps = [int(x) for x in a.checkpoints.split(",")]
    torch.set_num_threads(a.threads)
    train, held = datasets()
    ref = om.TriCoLoRef(om.BiGRURef(syn.DEFAULT_VOCAB, 512), om.MVCNNRef(512, 512, "resnet18", NV), om.SparseCNNRef(V, 32, 512, 512))
    fill_module(ref)
    opt = torch.optim.Adam(ref.parameters(), lr=3.5e-4, weight_decay=1e-6)        # config/config.yaml:50-53
    real = ropt = None
    if not a.restated_only:
        from oracle import make_golden as mg
        mg.install_shims()
        real, _ = mg._ref_net("BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", V, NV, S)     # the REAL TriCoLoNet, recipe weights
        sd_a, sd_b = real.state_dict(), ref.state_dict()
        assert sd_a.keys() == sd_b.keys() and all(torch.equal(sd_a[k], sd_b[k]) for k in sd_a), "reference and restatement must start identical"
        ropt = real.configure_optimizers()                                        # tricolo_net.py:43-44
    rng = np.random.default_rng(SEED_ORDER)
    out = {"steps": np.int64(a.steps), "checkpoints": np.array(cps), "train_sha": data_sha(train), "held_sha": data_sha(held),
           "source": np.array("reference" if real is not None else "restatement")}
    losses, rlosses, t0 = [], [], time.time()
    for step in range(1, a.steps + 1):
        batch = syn.collate_items([train[i] for i in batch_indices(rng, TRAIN_SHAPES)], voxel=True, views=True)
        opt.zero_grad(set_to_none=True)
        loss, _, _ = ref.training_step(batch)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        if real is not None:
            ropt.zero_grad(set_to_none=True)
            rl = real.training_step(batch, step)                                   # tricolo_net.py:67-71
            rl.backward()
            ropt.step()
            rlosses.append(float(rl))
        if step % 50 == 0:
            extra = f"  reference {rlosses[-1]:.4f}" if real is not None else ""
            print(f"step {step} loss {losses[-1]:.4f}{extra} ({time.time() - t0:.0f} s)", flush=True)
        if step in cps:
            ref.eval()
            m = metrics(embed(ref, held), held)
            ref.train()
            pre = "" if real is None else "restated/"
            out[f"{pre}cp{step}/recall_rate"] = m["recall_rate"]
            out[f"{pre}cp{step}/mrr"] = np.float64(m["mrr"])
            out[f"{pre}cp{step}/indices"] = m["indices"].astype(np.int16)
            if real is None:
                out[f"cp{step}/ndcg"] = m["ndcg"]
                out[f"cp{step}/labels"] = m["labels"].astype(np.int16)
            msg = f"checkpoint {step}: restatement RR@1 {100 * m['recall_rate'][0]:.2f} RR@5 {100 * m['recall_rate'][4]:.2f}"
            if real is not None:
                real.eval()
                rm = real_metrics(embed(real, held), held)
                real.train()
                out[f"cp{step}/recall_rate"] = rm["recall_rate"]
                out[f"cp{step}/ndcg"] = rm["ndcg"]
                out[f"cp{step}/mrr"] = np.float64(rm["mrr"])
                out[f"cp{step}/indices"] = rm["indices"].astype(np.int16)
                out[f"cp{step}/labels"] = rm["labels"].astype(np.int16)
                msg += f" | REFERENCE RR@1 {100 * rm['recall_rate'][0]:.2f} RR@5 {100 * rm['recall_rate'][4]:.2f}"
            print(msg, flush=True)
    if real is not None:
        out["losses"] = np.array(rlosses, dtype=np.float32)
        out["restated/losses"] = np.array(losses, dtype=np.float32)
    else:
        out["losses"] = np.array(losses, dtype=np.float32)
    np.savez_compressed(a.out, **out)


if __name__ == "__main__":
    main()
