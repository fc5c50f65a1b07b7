"""Restatement of the reference encoder / loss / container classes in plain PyTorch-CPU.  TEST INFRASTRUCTURE.

Each class cites the reference file:line it follows.  State-dict key names equal the reference's so the same
deterministic weight recipe (oracle/recipe.py) can be loaded into the reference class, this restatement and the
HIP-backed product modules.
"""
from itertools import combinations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import spconv_dense as spconv
from .resnet18 import resnet18


class SparseCNNRef(nn.Module):
    """/root/reference/tricolo/model/module/voxel_encoder/sparse_cnn.py:8-51.
    Deliberate generalisation: mlp[0] in_features = z_dim*(voxel_size//32)**3 instead of the
    hard-coded 4096 (identical at voxel_size=64; needed for the 32^3 configs of BASELINE.json)."""

    def __init__(self, voxel_size, ef_dim, z_dim, out_dim, **kwargs):
        super().__init__()
        self.voxel_size = voxel_size
        chans = [3, ef_dim, ef_dim * 2, ef_dim * 4, ef_dim * 8, z_dim]
        layers = []
        for i in range(5):                                                  # sparse_cnn.py:12-35
            layers += [spconv.SubMConv3d(chans[i], chans[i + 1], kernel_size=3, bias=False),
                       nn.BatchNorm1d(chans[i + 1]), nn.ReLU(inplace=True), spconv.SparseMaxPool3d(2, 2)]
        layers.append(spconv.ToDense())                                     # sparse_cnn.py:36
        self.sparseModel = spconv.SparseSequential(*layers)
        self.mlp = nn.Sequential(nn.Linear(z_dim * (voxel_size // 32) ** 3, out_dim), nn.ReLU(inplace=True),
                                 nn.Linear(out_dim, out_dim))              # sparse_cnn.py:39-44

    def forward(self, x, batch_size):
        t = spconv.SparseConvTensor(x["feats"], x["locs"], [self.voxel_size] * 3, batch_size)   # :47
        t = self.sparseModel(t)                                                                  # :48
        return F.normalize(self.mlp(t.reshape(t.shape[0], -1)), dim=1)                           # :49-51


class MVCNNRef(nn.Module):
    """/root/reference/tricolo/model/module/img_encoder/mv_cnn.py:13-33 (resnet18 branch of SVCNN :40-45)."""

    def __init__(self, z_dim, out_dim, cnn_name, num_views, **kwargs):
        super().__init__()
        assert cnn_name == "resnet18"
        net = resnet18()
        net.fc = nn.Linear(512, z_dim)                                     # mv_cnn.py:45
        self.num_views = num_views
        self.net_1 = nn.Sequential(*list(net.children())[:-1])            # mv_cnn.py:20
        self.net_2 = net.fc                                                # mv_cnn.py:21
        self.mlp = nn.Sequential(nn.Linear(z_dim, out_dim), nn.ReLU(inplace=True), nn.Linear(out_dim, out_dim))

    def forward(self, x, data_dict=None):
        y = self.net_1(x)                                                  # mv_cnn.py:29
        y = y.view((x.shape[0] // self.num_views, self.num_views, y.shape[-3], y.shape[-2], y.shape[-1]))
        y = self.net_2(torch.max(y, 1)[0].view(y.shape[0], -1))           # mv_cnn.py:31
        return F.normalize(self.mlp(y), dim=1)                             # mv_cnn.py:33


class BiGRURef(nn.Module):
    """/root/reference/tricolo/model/module/text_encoder/bigru.py:8-18.  No packing: pad tokens are stepped through."""

    def __init__(self, vocab_size, out_dim, **kwargs):
        super().__init__()
        self.embedding_layer = nn.Embedding(vocab_size, 256, padding_idx=0)
        self.gru = nn.GRU(input_size=256, hidden_size=128, num_layers=1, bidirectional=True)
        self.fc = nn.Linear(256, out_dim)

    def forward(self, x, data_dict=None):
        emb = torch.transpose(self.embedding_layer(x), 0, 1)
        h0 = torch.zeros((2, emb.shape[1], 128), dtype=torch.float32, device=emb.device)
        _, hidden = self.gru(emb, h0)
        return F.normalize(torch.tanh(self.fc(torch.cat((hidden[-2], hidden[-1]), dim=1))), dim=1)


def gru_explicit(emb, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """Published GRU cell equations (torch.nn.GRU docs), gate order (r, z, n); emb [L,B,I] -> final hidden [B,H].
    Used to cross-check nn.GRU and as the spec of the fused recurrence kernel."""
    L, B, _ = emb.shape
    H = w_hh.shape[1]
    h = emb.new_zeros((B, H))
    steps = range(L - 1, -1, -1) if reverse else range(L)
    hs = []
    for t in steps:
        gi = emb[t] @ w_ih.t() + b_ih
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        hs.append(h)
    return h, hs


class CLIPTextRef(nn.Module):
    """/root/reference/tricolo/model/module/text_encoder/clip_text.py:6-22 (clip_model only supplies output_dim)."""

    def __init__(self, out_dim, clip_model=None, clip_dim=768, **kwargs):
        super().__init__()
        in_dim = clip_model.visual.output_dim if clip_model is not None else clip_dim
        self.mlp = nn.Sequential(nn.Linear(in_dim, out_dim), nn.ReLU(inplace=True), nn.Dropout(0.1),
                                 nn.Linear(out_dim, out_dim))

    def forward(self, tokens, data_dict):
        return self.mlp(data_dict["clip_embeddings_text"])


def nt_xent_ref(zis, zjs, temperature, alpha_weight, norm=True):
    """/root/reference/tricolo/loss/nt_xent.py:24-74, with _softXEnt :15-22 written out."""
    if norm:
        zis = F.normalize(zis, p=2, dim=1)                                 # :56
        zjs = F.normalize(zjs, p=2, dim=1)                                 # :57
    B = zis.shape[0]
    logits_ab = zis @ zjs.t() / temperature                                # :68
    logits_ba = zjs @ zis.t() / temperature                                # :69
    eye = torch.eye(B, dtype=torch.float32, device=zis.device)             # :62
    loss_a = -(eye * F.log_softmax(logits_ab, dim=1)).sum() / B            # :20-21, :71
    loss_b = -(eye * F.log_softmax(logits_ba, dim=1)).sum() / B            # :72
    return alpha_weight * loss_a + (1 - alpha_weight) * loss_b             # :74


def nt_xent_numpy(za, zb, temperature, alpha_weight):
    """float64 numpy form of the same loss and its gradient w.r.t. the *normalised* inputs."""
    za = np.asarray(za, np.float64)
    zb = np.asarray(zb, np.float64)
    na = np.maximum(np.linalg.norm(za, axis=1, keepdims=True), 1e-12)
    nb = np.maximum(np.linalg.norm(zb, axis=1, keepdims=True), 1e-12)
    a, b = za / na, zb / nb
    S = a @ b.T / temperature
    B = S.shape[0]

    def lse(x, axis):
        m = x.max(axis=axis, keepdims=True)
        return (m + np.log(np.exp(x - m).sum(axis=axis, keepdims=True))).squeeze(axis)
    row, col = lse(S, 1), lse(S, 0)
    d = np.diag(S)
    loss = alpha_weight * (-(d - row).mean()) + (1 - alpha_weight) * (-(d - col).mean())
    P_row = np.exp(S - row[:, None])
    P_col = np.exp(S - col[None, :])
    dS = (alpha_weight * (P_row - np.eye(B)) + (1 - alpha_weight) * (P_col - np.eye(B))) / B
    da_hat = dS @ b / temperature
    db_hat = dS.T @ a / temperature
    da = (da_hat - a * (da_hat * a).sum(1, keepdims=True)) / na
    db = (db_hat - b * (db_hat * b).sum(1, keepdims=True)) / nb
    return loss, da, db


class TriCoLoRef(nn.Module):
    """/root/reference/tricolo/model/tricolo_net.py:46-71 (forward, _calculate_losses, training_step) without
    Lightning / Hydra: encoders and loss hyper-parameters are passed in."""

    def __init__(self, text_encoder, image_encoder=None, voxel_encoder=None, temperature=0.1, alpha_weight=0.25):
        super().__init__()
        self.text_encoder, self.image_encoder, self.voxel_encoder = text_encoder, image_encoder, voxel_encoder
        self.temperature, self.alpha_weight = temperature, alpha_weight

    def forward(self, data_dict):
        out = {"text_features": self.text_encoder(data_dict["tokens"], data_dict)}                  # :47-49
        if self.image_encoder is not None:
            out["image_features"] = self.image_encoder(data_dict["images"].flatten(end_dim=1), data_dict)   # :51
        if self.voxel_encoder is not None:
            out["voxel_features"] = self.voxel_encoder(data_dict["voxels"], len(data_dict["model_id"]))     # :53
        return out

    def calculate_losses(self, output_dict, loss_prefix):
        loss_dict = {}
        for a, b in combinations(output_dict.keys(), 2):                                            # :59-63
            loss_dict[f"{loss_prefix}/{a[:-9]}_{b[:-9]}_loss"] = nt_xent_ref(
                output_dict[a], output_dict[b], self.temperature, self.alpha_weight)
        loss_dict[f"{loss_prefix}/total_loss"] = sum(loss_dict.values())                           # :64
        return loss_dict

    def training_step(self, data_dict):
        out = self(data_dict)
        losses = self.calculate_losses(out, "train_loss")
        return losses["train_loss/total_loss"], losses, out


def adam_step_explicit(p, g, m, v, step, lr=3.5e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-6):
    """torch.optim.Adam single-tensor update (L2-in-gradient, NOT AdamW) as instantiated by
    /root/reference/config/config.yaml:50-53 + tricolo_net.py:43-44.  ``step`` is 1-based.  In place on p, m, v."""
    g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p
