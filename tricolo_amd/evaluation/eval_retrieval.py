"""Text -> shape retrieval metrics (RR@k, NDCG@k, MRR) - mirrors the public surface of
/root/reference/tricolo/evaluation/eval_retrieval.py:249-278 (`compute_metrics(dataset, embeddings_dict,
print_results=False) -> dict`) with the same quirks: the shape matrix is built from the FIRST occurrence of each
model_id (:49-56), similarities are plain dot products on a float64 text matrix (:24,:74), ranking is numpy's
ascending argsort flipped (:75-82, which fixes the tie order), RR@k counts queries with a correct shape in the top k.

The ranking itself runs on the GPU: `tri_retrieval_topk` computes the float64 similarities, the five
best shapes per query and the rank of the query's own shape in one launch (csrc/retrieval.hip); the host only builds the
label tables from the model_id strings and turns [Nq,5] indices + [Nq] ranks into the four numbers.  There is no host
ranking path in this package: without the HIP library / a GPU the call raises (the reference's numpy algorithm lives in
oracle/retrieval.py, test infrastructure).  Unlike the reference it does not write nearest.jsonl into the CWD unless asked.
"""
import json

import numpy as np


def construct_embeddings_matrix(dataset, embeddings_dict):
    tuples = embeddings_dict["caption_embedding_tuples"]
    dim = tuples[0][-1].shape[0]
    text = np.zeros((len(tuples), dim))
    labels = np.zeros(len(tuples), dtype=np.int64)
    model_id_to_label, label_to_model_id, shapes = {}, {}, []
    for idx, (_caption, category, model_id, text_emb, shape_emb) in enumerate(tuples):
        if dataset == "Primitives":
            model_id = category
        if model_id not in model_id_to_label:
            model_id_to_label[model_id] = len(shapes)
            label_to_model_id[len(shapes)] = model_id
            shapes.append(shape_emb)
        text[idx] = text_emb
        labels[idx] = model_id_to_label[model_id]
    shape = np.vstack(shapes)
    return text, shape, labels, np.arange(len(shapes)), model_id_to_label, len(tuples), label_to_model_id


def nearest_neighbors_hip(shape, text, labels, n_neighbors):
    """Device path: (distances [Nq,k] f64, indices [Nq,k], first_hit [Nq]) from csrc/retrieval.hip."""
    import torch
    from .. import ops
    if not torch.cuda.is_available():
        raise RuntimeError("compute_metrics ranks on the GPU (tri_retrieval_topk) and there is no CPU fallback: no MI355X visible")
    dev = torch.device("cuda")
    t = torch.from_numpy(np.ascontiguousarray(text, dtype=np.float32)).to(dev)     # the f64 text matrix holds f32 values
    s = torch.from_numpy(np.ascontiguousarray(shape, dtype=np.float32)).to(dev)
    lab = torch.from_numpy(np.ascontiguousarray(labels, dtype=np.int32)).to(dev)
    idx, sim, hit = ops.retrieval_topk(t, s, lab, n_neighbors)
    return sim.cpu().numpy(), idx.cpu().numpy().astype(np.int64), hit.cpu().numpy().astype(np.int64)


def compute_pr_at_k(indices, first_hit, labels, n_neighbors, num_embeddings, fit_labels):
    nearest = fit_labels[indices]
    rel = (nearest == labels[:, None]).astype(np.float32)
    num_correct = np.cumsum(rel, axis=1)
    num_relevant = np.bincount(fit_labels)[labels]
    rel_ideal = (np.arange(n_neighbors)[None, :] < np.minimum(num_relevant, n_neighbors)[:, None]).astype(np.float32)
    mrr = float(np.mean(1.0 / (first_hit + 1)))
    dcg_d = np.log2(np.arange(1, n_neighbors + 1) + 1)
    dcg = np.cumsum((np.exp2(rel) - 1) / dcg_d, axis=1)
    dcg_ideal = np.cumsum((np.exp2(rel_ideal) - 1) / dcg_d, axis=1)
    return {
        "precision": np.sum(num_correct / np.arange(1, n_neighbors + 1), axis=0) / num_embeddings,
        "recall": np.sum(num_correct / num_relevant[:, None], axis=0) / num_embeddings,
        "recall_rate": np.sum(num_correct > 0, axis=0) / num_embeddings,
        "ndcg": np.sum(dcg / dcg_ideal, axis=0) / num_embeddings,
        "mrr": mrr,
    }


def compute_metrics(dataset, embeddings_dict, print_results=False, nearest_path=None):
    text, shape, labels, fit_labels, _, num, label_to_model_id = construct_embeddings_matrix(dataset, embeddings_dict)
    n_neighbors = 5
    distances, indices, first_hit = nearest_neighbors_hip(shape, text, labels, n_neighbors)
    pr_at_k = compute_pr_at_k(indices, first_hit, labels, n_neighbors, num, fit_labels)
    pr_at_k["indices"] = indices
    if nearest_path is not None:
        tuples = embeddings_dict["caption_embedding_tuples"]
        with open(nearest_path, "w") as f:
            for i in range(num):
                f.write(json.dumps({"cat_id": tuples[i][1], "groundtruth": f"{tuples[i][2]}-{i:04d}",
                                    "retrieved_models": [label_to_model_id[int(c)] for c in indices[i]],
                                    "distance": distances[i].tolist()}) + "\n")
    if print_results:
        print("\nRR@1 RR@5 NDCG@5 MRR")
        print(f'{round(pr_at_k["recall_rate"][0] * 100, 2)} {round(pr_at_k["recall_rate"][4] * 100, 2)} '
              f'{round(pr_at_k["ndcg"][4] * 100, 2)} {round(pr_at_k["mrr"] * 100, 2)}')
    return pr_at_k
