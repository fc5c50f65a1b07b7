"""Generate tests/golden/*.npz by running the REAL reference classes from /root/reference.  TEST INFRASTRUCTURE.

Runs only in the build container (the reference never travels to the GPU box; the fixtures do).  The reference's
third-party imports that are not installed are replaced by in-memory shims:
  lightning.pytorch -> nn.Module + .device/.hparams/.log_dict     hydra       -> tricolo_amd.config.instantiate
  spconv.pytorch    -> oracle.spconv_dense (dense-masked)         torchvision -> oracle.resnet18
  clip / efficientnet_pytorch / jsonlines -> inert stubs
Everything else (module wiring, NTXentLoss, BiGRUEncoder, CLIPTextEncoder, TriCoLoNet.forward/_calculate_losses/
training_step/configure_optimizers, compute_metrics) executes the reference's own code.

    python -m oracle.make_golden          # rewrites tests/golden/
"""
import hashlib
import inspect
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import resnet18 as _resnet, spconv_dense as _spconv      # noqa: E402
from oracle.recipe import fill_module, probe                         # noqa: E402
from tricolo_amd import config as tcfg                                # noqa: E402
from tricolo_amd.data import synthetic as syn                         # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


# ---------------------------------------------------------------------------------------------- shims
def install_shims():
    class LightningModule(nn.Module):
        @property
        def device(self):
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        def save_hyperparameters(self):
            frame = inspect.currentframe().f_back
            self.hparams = types.SimpleNamespace(**{k: v for k, v in frame.f_locals.items()
                                                    if k not in ("self", "__class__")})

        def log_dict(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        def print(self, *a, **k):
            pass

    pl = types.ModuleType("lightning.pytorch")
    pl.LightningModule = LightningModule
    pl.LightningDataModule = object
    lightning = types.ModuleType("lightning")
    lightning.pytorch = pl
    sys.modules.update({"lightning": lightning, "lightning.pytorch": pl})

    hydra = types.ModuleType("hydra")
    hydra.utils = types.ModuleType("hydra.utils")
    hydra.utils.instantiate = tcfg.instantiate
    hydra.main = lambda **kw: (lambda f: f)
    sys.modules.update({"hydra": hydra, "hydra.utils": hydra.utils})

    spconv = types.ModuleType("spconv")
    spconv.pytorch = _spconv
    sys.modules.update({"spconv": spconv, "spconv.pytorch": _spconv})

    tv = types.ModuleType("torchvision")
    tv.models = _resnet
    sys.modules.update({"torchvision": tv, "torchvision.models": _resnet})

    eff = types.ModuleType("efficientnet_pytorch")
    eff.EfficientNet = type("EfficientNet", (), {})
    sys.modules["efficientnet_pytorch"] = eff

    clip = types.ModuleType("clip")

    class _Clip:
        visual = types.SimpleNamespace(output_dim=768)

        def parameters(self):
            return []
    clip.load = lambda name, device=None: (_Clip(), None)
    sys.modules["clip"] = clip

    jl = types.ModuleType("jsonlines")

    class _W:
        def write(self, obj):
            pass
    jl.open = lambda path, mode="r": _W()
    sys.modules["jsonlines"] = jl
    if REF not in sys.path:
        sys.path.insert(0, REF)


def sha(*tensors) -> str:
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.numpy() if isinstance(t, torch.Tensor) else t).tobytes())
    return h.hexdigest()


def batch_sha(batch) -> str:
    parts = [batch["tokens"]]
    if "images" in batch:
        parts.append(batch["images"])
    if "voxels" in batch:
        parts += [batch["voxels"]["locs"], batch["voxels"]["feats"]]
    if "clip_embeddings_text" in batch:
        parts.append(batch["clip_embeddings_text"])
    return sha(*parts)


def grad_probes(module: nn.Module, out: dict, prefix: str):
    for name, p in module.named_parameters():
        if p.grad is None:
            continue
        n, s = probe(p.grad)
        out[f"{prefix}gradnorm/{name}"] = np.float64(n)
        out[f"{prefix}gradsample/{name}"] = s


def weight_probes(module: nn.Module, out: dict, prefix: str):
    for name, p in module.state_dict().items():
        if not p.dtype.is_floating_point:
            continue
        n, s = probe(p)
        out[f"{prefix}wnorm/{name}"] = np.float64(n)
        out[f"{prefix}wsample/{name}"] = s


# ---------------------------------------------------------------------------------------------- cases
def golden_ntxent():
    from tricolo.loss.nt_xent import NTXentLoss
    out = {}
    for tag, B, T, alpha in (("b8", 8, 0.1, 0.25), ("b5", 5, 0.1, 0.25), ("b16_sym", 16, 0.07, 0.5), ("b1", 1, 0.1, 0.25)):
        g = torch.Generator().manual_seed(1000 + B)
        za = torch.randn(B, 512, generator=g).requires_grad_()
        zb = (torch.randn(B, 512, generator=g) * 3).requires_grad_()
        loss = NTXentLoss(T, alpha)(za, zb)
        loss.backward()
        out.update({f"{tag}/za": za.detach().numpy(), f"{tag}/zb": zb.detach().numpy(),
                    f"{tag}/loss": np.float32(loss.item()), f"{tag}/dza": za.grad.numpy(), f"{tag}/dzb": zb.grad.numpy(),
                    f"{tag}/T": np.float32(T), f"{tag}/alpha": np.float32(alpha)})
        # alpha asymmetry: swapped arguments (SURVEY 8c viii)
        out[f"{tag}/loss_swapped"] = np.float32(NTXentLoss(T, alpha)(zb.detach(), za.detach()).item())
    np.savez_compressed(os.path.join(GOLD, "ntxent.npz"), **out)


def golden_bigru():
    from tricolo.model.module.text_encoder.bigru import BiGRUEncoder
    m = BiGRUEncoder(vocab_size=syn.DEFAULT_VOCAB, out_dim=512)
    fill_module(m, prefix="text_encoder.")
    batch = syn.make_batch(8, voxel_size=None, num_views=None, seed=syn.BASE_SEED + 6)
    z = m(batch["tokens"], batch)
    g = torch.Generator().manual_seed(7)
    up = torch.randn(z.shape, generator=g)
    (z * up).sum().backward()
    out = {"input_sha": sha(batch["tokens"]), "z": z.detach().numpy(), "upstream": up.numpy()}
    grad_probes(m, out, "")
    # all-pad sequence (SURVEY 8c ix)
    zpad = m(torch.zeros((2, 96), dtype=torch.int32), {})
    out["z_allpad"] = zpad.detach().numpy()
    np.savez_compressed(os.path.join(GOLD, "bigru.npz"), **out)


def golden_clip_text():
    from tricolo.model.module.text_encoder.clip_text import CLIPTextEncoder
    import clip
    m = CLIPTextEncoder(out_dim=512, clip_model=clip.load("ViT-L/14")[0])
    fill_module(m, prefix="text_encoder.")
    m.eval()                                      # Dropout(0.1) off: deterministic comparison
    batch = syn.make_batch(8, voxel_size=None, num_views=None, clip_text=True, seed=syn.BASE_SEED + 5)
    z = m(batch["tokens"], batch)
    np.savez_compressed(os.path.join(GOLD, "clip_text.npz"), z=z.detach().numpy(),
                        input_sha=sha(batch["clip_embeddings_text"]))


def _ref_voxel_encoder(V):
    from tricolo.model.module.voxel_encoder.sparse_cnn import SparseCNNEncoder
    m = SparseCNNEncoder(voxel_size=V, ef_dim=32, z_dim=512, out_dim=512)
    if V != 64:                                   # the one deliberate deviation, SURVEY section 0.2
        m.mlp[0] = nn.Linear(512 * (V // 32) ** 3, 512)
    return m


def golden_voxel():
    out = {}
    for tag, V, B in (("v32", 32, 8), ("v64", 64, 2)):
        m = _ref_voxel_encoder(V)
        fill_module(m, prefix="voxel_encoder.")
        batch = syn.make_batch(B, voxel_size=V, num_views=None, seed=syn.BASE_SEED + (1 if V == 32 else 8))
        z = m(batch["voxels"], B)
        g = torch.Generator().manual_seed(11)
        up = torch.randn(z.shape, generator=g)
        (z * up).sum().backward()
        out.update({f"{tag}/input_sha": sha(batch["voxels"]["locs"], batch["voxels"]["feats"]),
                    f"{tag}/n_active": np.int64(batch["voxels"]["locs"].shape[0]),
                    f"{tag}/z": z.detach().numpy(), f"{tag}/upstream": up.numpy()})
        grad_probes(m, out, f"{tag}/")
        weight_probes(m, out, f"{tag}/after/")     # running stats after one train-mode forward
    # edge: one empty sample inside a batch (SURVEY 8c vi)
    m = _ref_voxel_encoder(32)
    fill_module(m, prefix="voxel_encoder.")
    batch = syn.make_batch(3, voxel_size=32, num_views=None, seed=syn.BASE_SEED + 9)
    keep = batch["voxels"]["locs"][:, 0] != 1
    vox = {"locs": batch["voxels"]["locs"][keep], "feats": batch["voxels"]["feats"][keep]}
    out["empty1/z"] = m(vox, 3).detach().numpy()
    np.savez_compressed(os.path.join(GOLD, "voxel.npz"), **out)


def golden_mvcnn():
    from tricolo.model.module.img_encoder.mv_cnn import MVCNNEncoder
    out = {}
    for tag, B, nv, S in (("v6s128", 8, 6, 128), ("v12s224", 2, 12, 224)):
        m = MVCNNEncoder(z_dim=512, out_dim=512, cnn_name="resnet18", num_views=nv)
        fill_module(m, prefix="image_encoder.")
        batch = syn.make_batch(B, voxel_size=None, num_views=nv, image_size=S, seed=syn.BASE_SEED + 3)
        z = m(batch["images"].flatten(end_dim=1), batch)
        g = torch.Generator().manual_seed(13)
        up = torch.randn(z.shape, generator=g)
        (z * up).sum().backward()
        out.update({f"{tag}/input_sha": sha(batch["images"]), f"{tag}/z": z.detach().numpy(),
                    f"{tag}/upstream": up.numpy()})
        grad_probes(m, out, f"{tag}/")
        weight_probes(m, out, f"{tag}/after/")
    np.savez_compressed(os.path.join(GOLD, "mvcnn.npz"), **out)


def _ref_net(text, image, voxel, V, nv, S):
    from tricolo.model.tricolo_net import TriCoLoNet
    ov = [f"model.text_encoder={text}", f"model.image_encoder={image or 'null'}",
          f"model.voxel_encoder={voxel or 'null'}", "data=text2shape_chair_table", f"data.voxel_size={V}",
          f"data.num_views={nv}", f"data.image_size={S}", "experiment_name=golden"]
    cfg = tcfg.compose(os.path.join(REF, "config"), "config", ov)
    net = TriCoLoNet(cfg)
    if voxel and V != 64:
        net.voxel_encoder.mlp[0] = nn.Linear(512 * (V // 32) ** 3, 512)
    fill_module(net)
    if text == "CLIPTextEncoder":
        net.text_encoder.mlp[2].eval()            # Dropout off
    return net, cfg


def golden_steps():
    cases = (
        ("cfg1_biV", "BiGRUEncoder", None, "SparseCNNEncoder", 32, None, 128, 8, 1, False),
        ("cfg3_biI", "BiGRUEncoder", "MVCNNEncoder", None, 32, 6, 128, 8, 3, False),
        ("cfg4_tri", "BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", 32, 6, 128, 8, 4, False),
        ("cfg5_tri64", "CLIPTextEncoder", "MVCNNEncoder", "SparseCNNEncoder", 64, 12, 224, 2, 5, True),
    )
    for tag, text, image, voxel, V, nv, S, B, seed_off, clip_text in cases:
        net, cfg = _ref_net(text, image, voxel, V, nv or 6, S)
        batch = syn.make_batch(B, voxel_size=V if voxel else None, num_views=nv if image else None, image_size=S,
                               clip_text=clip_text, seed=syn.BASE_SEED + seed_off)
        out = {"input_sha": batch_sha(batch), "B": np.int64(B)}
        opt = net.configure_optimizers()              # tricolo_net.py:43-44 -> torch.optim.Adam(lr, weight_decay)
        for step in range(4):
            opt.zero_grad(set_to_none=True)
            emb = net(batch)
            for k in emb:
                emb[k].retain_grad()
            losses = net._calculate_losses(emb, "train_loss")
            total = losses["train_loss/total_loss"]
            out[f"step{step}/total_loss"] = np.float32(total.item())
            for k, v in losses.items():
                out[f"step{step}/{k}"] = np.float32(v.item())
            if step == 3:
                break
            total.backward()
            if step == 0:
                for k, v in emb.items():
                    out[f"emb/{k}"] = v.detach().numpy()
                    out[f"demb/{k}"] = v.grad.numpy()
                grad_probes(net, out, "")
            opt.step()
        weight_probes(net, out, "after3/")
        np.savez_compressed(os.path.join(GOLD, f"step_{tag}.npz"), **out)
        print(tag, {k: float(v) for k, v in out.items() if k.endswith("total_loss")})


def golden_dp2():
    """Two data-parallel shards of 4: per-shard encoders (local BatchNorm statistics, the Lightning-DDP default)
    + NT-Xent over the gathered global batch.  Gradients = d(global loss)/d(theta)."""
    from tricolo.loss.nt_xent import NTXentLoss
    from itertools import combinations
    net, cfg = _ref_net("BiGRUEncoder", "MVCNNEncoder", "SparseCNNEncoder", 32, 6, 128)
    full = syn.make_batch(8, voxel_size=32, num_views=6, image_size=128, seed=syn.BASE_SEED + 4)
    embs = []
    for r in range(2):
        sl = slice(4 * r, 4 * r + 4)
        keep = (full["voxels"]["locs"][:, 0] >= 4 * r) & (full["voxels"]["locs"][:, 0] < 4 * r + 4)
        locs = full["voxels"]["locs"][keep].clone()
        locs[:, 0] -= 4 * r
        shard = {"model_id": full["model_id"][sl], "category": full["category"][sl], "tokens": full["tokens"][sl],
                 "images": full["images"][sl], "voxels": {"locs": locs, "feats": full["voxels"]["feats"][keep]}}
        embs.append(net(shard))
    glob = {k: torch.cat([e[k] for e in embs]) for k in embs[0]}
    loss_fn = NTXentLoss(cfg.loss.NTXentLoss.temperature, cfg.loss.NTXentLoss.alpha_weight)
    out = {"input_sha": batch_sha(full)}
    total = 0
    for a, b in combinations(glob.keys(), 2):
        l = loss_fn(glob[a], glob[b])
        out[f"loss/{a[:-9]}_{b[:-9]}"] = np.float32(l.item())
        total = total + l
    out["total_loss"] = np.float32(total.item())
    total.backward()
    for k, v in glob.items():
        out[f"emb/{k}"] = v.detach().numpy()
    grad_probes(net, out, "")
    np.savez_compressed(os.path.join(GOLD, "dp2_tri.npz"), **out)


def golden_retrieval():
    from tricolo.evaluation import eval_retrieval as er
    rng = np.random.default_rng(4242)
    ns, cap, D = 64, 4, 64
    shape_lat = rng.standard_normal((ns, D)).astype(np.float32)
    model_ids, text, image, voxel = [], [], [], []
    for s in range(ns):
        for c in range(cap):
            model_ids.append(f"shape{s:04d}")
            text.append(shape_lat[s] + 2.5 * rng.standard_normal(D))
            image.append(shape_lat[s] * 0.5 + 0.1 * rng.standard_normal(D))
            voxel.append(shape_lat[s] * 0.5 + 0.1 * rng.standard_normal(D))
    perm = rng.permutation(len(model_ids))
    model_ids = [model_ids[i] for i in perm]
    text = np.asarray(text, np.float32)[perm]
    image = np.asarray(image, np.float32)[perm]
    voxel = np.asarray(voxel, np.float32)[perm]
    text /= np.linalg.norm(text, axis=1, keepdims=True)
    shape_feat = image + voxel                            # tricolo_net.py:134-138
    emb = {"caption_embedding_tuples": [(None, "synthetic", model_ids[i], text[i], shape_feat[i])
                                        for i in range(len(model_ids))]}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            np.random.seed(0)
            pr = er.compute_metrics("Text2ShapeChairTable", emb)
            (tm, sm, labels, fit_labels, _, _, _) = er.construct_embeddings_matrix("Text2ShapeChairTable", emb)
            _, indices, _ = er.compute_nearest_neighbors(sm, tm, 5)
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(GOLD, "retrieval.npz"), model_ids=np.asarray(model_ids), text=text, image=image,
                        voxel=voxel, recall_rate=pr["recall_rate"], ndcg=pr["ndcg"], mrr=np.float64(pr["mrr"]),
                        precision=pr["precision"], recall=pr["recall"], indices=indices, labels=labels)


def main():
    os.makedirs(GOLD, exist_ok=True)
    install_shims()
    torch.manual_seed(123)
    torch.set_num_threads(8)
    golden_ntxent()
    golden_bigru()
    golden_clip_text()
    golden_retrieval()
    golden_voxel()
    golden_mvcnn()
    golden_steps()
    golden_dp2()
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
