"""numpy restatement of the text->shape retrieval metric.  TEST INFRASTRUCTURE.

Follows /root/reference/tricolo/evaluation/eval_retrieval.py: construct_embeddings_matrix :6-65,
_compute_nearest_neighbors_cosine :68-81 (fit_eq_query=False branch; text and shape matrices never coincide),
compute_pr_at_k :149-207, compute_metrics :249-278, and tricolo_net.py:125-158 (_collate_output: shape embedding =
image + voxel features, summed, not re-normalised).  Side effects (nearest.jsonl, printing) are dropped.
"""
import numpy as np


def collate_shape_embedding(text, image=None, voxel=None):
    shape = np.zeros_like(text)                 # tricolo_net.py:134
    if image is not None:
        shape = shape + image                    # :135-136
    if voxel is not None:
        shape = shape + voxel                    # :137-138
    return shape


def compute_metrics_ref(model_ids, text_emb, shape_emb, n_neighbors=5):
    """model_ids: list[str] per caption; text_emb / shape_emb: [Nq, D] per caption (f32).
    Returns dict(recall_rate[5], ndcg[5], mrr, indices[Nq,5], fit_labels, labels)."""
    text = np.zeros((len(model_ids), text_emb.shape[1]))            # f64 matrix, eval_retrieval.py:24
    label_of, shape_rows, labels = {}, [], np.zeros(len(model_ids), dtype=np.int64)
    for i, mid in enumerate(model_ids):                              # first occurrence defines the shape row :49-56
        if mid not in label_of:
            label_of[mid] = len(shape_rows)
            shape_rows.append(shape_emb[i])
        text[i] = text_emb[i]
        labels[i] = label_of[mid]
    shape = np.vstack(shape_rows)                                    # keeps f32 dtype of the inputs (:62)
    fit_labels = np.arange(len(shape_rows))
    sims = np.dot(text, shape.T)                                     # :74
    sort_indices = np.argsort(sims, axis=1)                          # ascending :75
    indices = np.flip(sort_indices[:, -n_neighbors:], 1)             # :80-81
    sort_indices = np.flip(sort_indices, 1)                          # :82
    nq = len(model_ids)
    nearest_cls = fit_labels[indices]
    rel = (nearest_cls == labels[:, None]).astype(np.float32)        # :176
    num_correct = np.cumsum(rel, axis=1)                             # :179-182
    num_relevant = np.bincount(fit_labels)[labels]
    rel_ideal = np.zeros((nq, n_neighbors), dtype=np.float32)
    for i in range(nq):
        rel_ideal[i, :min(num_relevant[i], n_neighbors)] = 1         # :177
    all_cls = fit_labels[sort_indices]
    first_hit = np.argmax(all_cls == labels[:, None], axis=1)        # :185-187
    mrr = float(np.mean(1.0 / (first_hit + 1)))
    dcg_d = np.log2(np.arange(1, n_neighbors + 1) + 1)
    dcg = np.cumsum((np.exp2(rel) - 1) / dcg_d, axis=1)
    dcg_ideal = np.cumsum((np.exp2(rel_ideal) - 1) / dcg_d, axis=1)
    ndcg = np.sum(dcg / dcg_ideal, axis=0) / nq                      # :190-198
    recall_rate = np.sum(num_correct > 0, axis=0) / nq               # :199
    return {"recall_rate": recall_rate, "ndcg": ndcg, "mrr": mrr, "indices": indices,
            "labels": labels, "fit_labels": fit_labels}
