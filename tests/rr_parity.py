#!/usr/bin/env python
"""(Checker script, kept under tests/ because it drives the CPU oracle - test infrastructure - next to the HIP path;
not collected by pytest.)  Retrieval-quality parity on a learnable held-out synthetic set (north-star: RR@1 within +-0.2 of the reference).

Trains the HIP path (MI355X) and the CPU oracle (restatement of the reference step, pinned to the reference by
tests/golden) from IDENTICAL weights on IDENTICAL batches for a fixed number of steps, then embeds the same held-out
captions / shapes in eval mode with both, and reports RR@1 / RR@5 / NDCG@5 / MRR through the reference's metric
(eval_retrieval.compute_metrics semantics) plus the agreement of the top-1 retrieved indices.

    python tests/rr_parity.py [--steps 40] [--batch 32] [--train-shapes 256] [--eval-shapes 256] [--precision bf16x3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import modules as om  # noqa: E402
from oracle.retrieval import collate_shape_embedding, compute_metrics_ref  # noqa: E402
from tricolo_amd import config as tcfg, ops  # noqa: E402
from tricolo_amd.data import synthetic as syn  # noqa: E402
from tricolo_amd.model.tricolo_net import TriCoLoNet  # noqa: E402


def embed(net, items, bs, device, voxel, views):
    out = {"text": [], "image": [], "voxel": []}
    with torch.no_grad():
        for i in range(0, len(items), bs):
            b = syn.collate_items(items[i:i + bs], voxel=voxel, views=views)
            if device is not None:
                b = syn.batch_to_device(b, device)
            e = net(b)
            out["text"].append(e["text_features"].float().cpu().numpy())
            if "image_features" in e:
                out["image"].append(e["image_features"].float().cpu().numpy())
            if "voxel_features" in e:
                out["voxel"].append(e["voxel_features"].float().cpu().numpy())
    return {k: np.concatenate(v) for k, v in out.items() if v}


def run(args):
    dev = torch.device("cuda:0")
    ops.set_default_precision(args.precision)
    V, nv, S = args.voxel_size, args.num_views, args.image_size
    cfg = tcfg.compose(overrides=["data=synthetic", "model.text_encoder=BiGRUEncoder", "model.image_encoder=MVCNNEncoder",
                                  "model.voxel_encoder=SparseCNNEncoder", f"data.voxel_size={V}", f"data.num_views={nv}",
                                  f"data.image_size={S}", "experiment_name=rr"])
    torch.manual_seed(cfg.train_seed)
    net = TriCoLoNet(cfg)
    ref = om.TriCoLoRef(om.BiGRURef(syn.DEFAULT_VOCAB, 512), om.MVCNNRef(512, 512, "resnet18", nv), om.SparseCNNRef(V, 32, 512, 512))
    ref.load_state_dict(net.state_dict())
    net = net.to(dev)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    train = syn.make_retrieval_set(args.train_shapes, 2, V, nv, S, seed=syn.BASE_SEED + 99)
    held = syn.make_retrieval_set(args.eval_shapes, args.eval_captions, V, nv, S, seed=syn.BASE_SEED + 199)
    opt = net.configure_optimizers()
    ropt = torch.optim.Adam(ref.parameters(), lr=cfg.optimizer.lr, weight_decay=cfg.optimizer.weight_decay)
    rng = np.random.default_rng(7)
    log = []
    t0 = time.time()
    for step in range(args.steps):
        idx = rng.choice(len(train), size=args.batch, replace=False)
        # one caption per shape inside a batch (duplicate shapes would be false negatives for both implementations alike)
        batch = syn.collate_items([train[i] for i in idx], voxel=True, views=True)
        opt.zero_grad(set_to_none=True)
        loss = net.training_step(syn.batch_to_device(batch, dev), step)
        loss.backward()
        opt.step()
        ropt.zero_grad(set_to_none=True)
        rloss, _, _ = ref.training_step(batch)
        rloss.backward()
        ropt.step()
        log.append((float(loss.item()), float(rloss.item())))
        if step % 10 == 0 or step == args.steps - 1:
            print(f"step {step:3d}  hip loss {log[-1][0]:.5f}  oracle loss {log[-1][1]:.5f}  ({time.time() - t0:.0f} s)", flush=True)
    net.eval()
    ref.eval()
    if args.eval_on == "train":
        held = train                                       # seen captions / shapes: retrieval is well above chance here
    e_hip = embed(net, held, args.batch, dev, True, True)
    e_ref = embed(ref, held, args.batch, None, True, True)
    ids = [f"shape{it['shape']:05d}" for it in held]
    m_hip = compute_metrics_ref(ids, e_hip["text"], collate_shape_embedding(e_hip["text"], e_hip.get("image"), e_hip.get("voxel")))
    m_ref = compute_metrics_ref(ids, e_ref["text"], collate_shape_embedding(e_ref["text"], e_ref.get("image"), e_ref.get("voxel")))
    top1_agree = float(np.mean(m_hip["indices"][:, 0] == m_ref["indices"][:, 0]))
    res = {
        "precision": args.precision, "eval_on": args.eval_on, "steps": args.steps, "batch": args.batch, "queries": len(held),
        "loss_first": log[0], "loss_last": log[-1], "max_loss_diff": max(abs(a - b) for a, b in log),
        "hip": {"RR@1": 100 * m_hip["recall_rate"][0], "RR@5": 100 * m_hip["recall_rate"][4], "NDCG@5": 100 * m_hip["ndcg"][4], "MRR": 100 * m_hip["mrr"]},
        "oracle": {"RR@1": 100 * m_ref["recall_rate"][0], "RR@5": 100 * m_ref["recall_rate"][4], "NDCG@5": 100 * m_ref["ndcg"][4], "MRR": 100 * m_ref["mrr"]},
        "top1_index_agreement": top1_agree,
        "max_embedding_diff": {k: float(np.abs(e_hip[k] - e_ref[k]).max()) for k in e_hip},
    }
    res["delta_RR@1"] = res["hip"]["RR@1"] - res["oracle"]["RR@1"]
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--train-shapes", type=int, default=256)
    ap.add_argument("--eval-shapes", type=int, default=256)
    ap.add_argument("--eval-captions", type=int, default=5)
    ap.add_argument("--voxel-size", type=int, default=32)
    ap.add_argument("--num-views", type=int, default=6)
    ap.add_argument("--image-size", type=int, default=128)
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--eval-on", default="heldout", choices=["heldout", "train"])
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    r = run(a)
    print(json.dumps(r, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(r, f, indent=1)
