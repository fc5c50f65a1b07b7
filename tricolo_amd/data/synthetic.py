"""Deterministic synthetic batches with the layout of the reference's collate function.

The reference reads ShapeNet/Text2Shape npz + json from disk
(/root/reference/tricolo/data/dataset/general_dataset.py:12-98) and collates with
/root/reference/tricolo/data/data_module.py:40-65.  Neither dataset is available offline, so the
bench, the smoke test and the parity tests use this generator.  It reproduces the *conversions* of the
reference exactly (u8 RGBA grid -> COO locs/feats, u8 image -> CLIP-normalised f32, int tokens padded with 0)
on procedurally generated u8 inputs.  Only integer draws from ``numpy.random.default_rng`` are used so the
bytes are identical on every machine.
"""
from __future__ import annotations

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # general_dataset.py:87-89
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
MAX_TOKENS = 96
DEFAULT_VOCAB = 3588                              # config/data/text2shape_chair_table.yaml:15
BASE_SEED = 20250718


def _unit_grid(V):
    ax = np.arange(V, dtype=np.int32)
    return np.meshgrid(ax, ax, ax, indexing="ij")


def make_voxel_grid_u8(rng: np.random.Generator, V: int, attrs=None) -> np.ndarray:
    """One RGBA u8 grid [4,V,V,V]: union of 2-4 boxes / ellipsoids, 8-20 % filled, alpha in {0,255}."""
    zz, yy, xx = _unit_grid(V)
    for _attempt in range(64):
        occ = np.zeros((V, V, V), dtype=bool)
        rgb = np.zeros((3, V, V, V), dtype=np.int32)
        nparts = int(rng.integers(2, 5)) if attrs is None else int(attrs["nparts"])
        for p in range(nparts):
            if attrs is None:
                half = rng.integers(max(2, V // 10), max(3, V // 3), size=3)
                cen = rng.integers(V // 4, V - V // 4, size=3)
                kind = int(rng.integers(0, 2))
                base = rng.integers(24, 232, size=3)
            else:
                half, cen, kind, base = attrs["half"][p], attrs["cen"][p], attrs["kind"][p], attrs["base"][p]
            if kind == 0:
                part = ((np.abs(zz - cen[0]) <= half[0]) & (np.abs(yy - cen[1]) <= half[1])
                        & (np.abs(xx - cen[2]) <= half[2]))
            else:
                # integer ellipsoid test: sum((d*prod_other)^2) <= prod_all^2
                h = half.astype(np.int64)
                d0 = (zz - cen[0]).astype(np.int64) * h[1] * h[2]
                d1 = (yy - cen[1]).astype(np.int64) * h[0] * h[2]
                d2 = (xx - cen[2]).astype(np.int64) * h[0] * h[1]
                part = (d0 * d0 + d1 * d1 + d2 * d2) <= (h[0] * h[1] * h[2]) ** 2
            noise = rng.integers(-16, 17, size=(3, V, V, V))
            for c in range(3):
                rgb[c][part] = np.clip(int(base[c]) + noise[c][part], 0, 255)
            occ |= part
        frac = occ.mean()
        if attrs is not None or 0.08 <= frac <= 0.20:
            break
    grid = np.zeros((4, V, V, V), dtype=np.uint8)
    grid[:3] = rgb.astype(np.uint8) * occ[None]
    grid[3] = occ.astype(np.uint8) * 255
    return grid


def grid_to_sparse(grid_u8: np.ndarray):
    """RGBA u8 [4,V,V,V] -> (coords i32 [n,3], feats f32 [n,3]); mirrors general_dataset.py:47-51,92-93."""
    grid = np.transpose(grid_u8, (1, 2, 3, 0))
    flat = grid.reshape(-1, grid.shape[3])
    solid = flat[:, -1].nonzero()
    coords = (np.indices(grid.shape[:3], dtype=np.uint8).reshape(3, -1).T)[solid]
    feats = flat[:, :3][solid]
    return coords.astype(np.int32), feats.astype(np.float32) / 255


def make_images_u8(rng: np.random.Generator, nviews: int, S: int, attrs=None) -> np.ndarray:
    """u8 [nviews,3,S,S]: low-frequency integer pattern + uniform noise."""
    yy, xx = np.meshgrid(np.arange(S, dtype=np.int64), np.arange(S, dtype=np.int64), indexing="ij")
    out = np.zeros((nviews, 3, S, S), dtype=np.uint8)
    for v in range(nviews):
        for c in range(3):
            if attrs is None:
                a, b, ph = (int(x) for x in rng.integers(1, 6, size=3))
                base = int(rng.integers(40, 200))
            else:
                a, b, ph, base = (int(x) for x in attrs["img"][v][c])
            # triangle waves (integer only) with period S/a and S/b
            pa, pb = max(2, S // a), max(2, S // b)
            tri = np.abs(((yy + ph * 3) % pa) * 2 - pa) * 48 // pa + np.abs(((xx + ph * 5) % pb) * 2 - pb) * 48 // pb
            noise = rng.integers(-12, 13, size=(S, S))
            out[v, c] = np.clip(base + tri - 48 + noise, 0, 255).astype(np.uint8)
    return out


def normalise_images(img_u8: np.ndarray) -> torch.Tensor:
    """u8 [...,3,S,S] -> f32 CLIP-normalised, as general_dataset.py:87-89 (Normalize(x/255))."""
    x = torch.from_numpy(img_u8).to(torch.float32) / 255
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=torch.float32).view(3, 1, 1)
    return (x - mean) / std


def make_tokens(rng: np.random.Generator, vocab: int = DEFAULT_VOCAB, min_len: int = 6, max_len: int = 64):
    n = int(rng.integers(min_len, max_len + 1))
    tok = np.zeros(MAX_TOKENS, dtype=np.int32)
    tok[:n] = rng.integers(1, vocab, size=n)
    return tok


def make_batch(batch_size: int, voxel_size: int | None = 32, num_views: int | None = 6, image_size: int = 128,
               vocab_size: int = DEFAULT_VOCAB, clip_text: bool = False, seed: int = BASE_SEED, rank: int = 0,
               keep_grids: bool = False) -> dict:
    """Batch dict with the keys of data_module.py:40-65 (`model_id`, `category`, `tokens`, `images`,
    `voxels`={'locs','feats'}, optional `clip_embeddings_text`).  Per-rank seed = seed*1000 + rank."""
    rng = np.random.default_rng(seed * 1000 + rank)
    d: dict = {"model_id": [f"syn{seed}_{rank}_{i:05d}" for i in range(batch_size)],
               "category": ["synthetic"] * batch_size}
    d["tokens"] = torch.from_numpy(np.stack([make_tokens(rng, vocab_size) for _ in range(batch_size)]))
    if num_views:
        imgs = np.stack([make_images_u8(rng, num_views, image_size) for _ in range(batch_size)])
        d["images"] = normalise_images(imgs)
    if voxel_size:
        locs, feats, grids = [], [], []
        for i in range(batch_size):
            g = make_voxel_grid_u8(rng, voxel_size)
            c, f = grid_to_sparse(g)
            # data_module.py:52-57: prepend in-batch sample index
            locs.append(np.concatenate([np.full((c.shape[0], 1), i, dtype=np.int32), c], axis=1))
            feats.append(f)
            if keep_grids:
                grids.append(g)
        d["voxels"] = {"locs": torch.from_numpy(np.concatenate(locs)), "feats": torch.from_numpy(np.concatenate(feats))}
        if keep_grids:
            d["voxel_grids_u8"] = torch.from_numpy(np.stack(grids))
    if clip_text:
        # integer draws -> float, unit-normalised (extract_clip_feats.py:30-31 normalises CLIP vectors)
        v = rng.integers(-1000, 1001, size=(batch_size, 768)).astype(np.float32)
        v /= np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-12)
        d["clip_embeddings_text"] = torch.from_numpy(v.astype(np.float32))
    return d


def batch_to_device(d: dict, device) -> dict:
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.to(device)
        elif isinstance(v, dict):
            out[k] = {kk: vv.to(device) for kk, vv in v.items()}
        else:
            out[k] = v
    return out


# ------------------------------------------------------------------------------------------------------
# Learnable retrieval set: each shape is drawn from a latent attribute vector that determines its voxels,
# its views and (noisily) its caption tokens, so contrastive training has signal.
# ------------------------------------------------------------------------------------------------------
def make_retrieval_set(num_shapes: int, captions_per_shape: int, voxel_size: int | None, num_views: int | None,
                       image_size: int, vocab_size: int = DEFAULT_VOCAB, seed: int = BASE_SEED + 99):
    rng = np.random.default_rng(seed)
    shapes = []
    for s in range(num_shapes):
        nparts = int(rng.integers(2, 5))
        V = voxel_size or 32
        attrs = {
            "nparts": nparts,
            "half": rng.integers(max(2, V // 10), max(3, V // 4), size=(nparts, 3)),
            "cen": rng.integers(V // 4, V - V // 4, size=(nparts, 3)),
            "kind": rng.integers(0, 2, size=nparts),
            "base": rng.integers(24, 232, size=(nparts, 3)),
        }
        nv = num_views or 1
        attrs["img"] = np.concatenate([rng.integers(1, 6, size=(nv, 3, 3)),
                                       rng.integers(40, 200, size=(nv, 3, 1))], axis=2)
        # attribute tokens: a deterministic function of the latent (quantised part sizes / colours)
        words = [1 + nparts]
        for p in range(nparts):
            words.append(16 + int(attrs["kind"][p]) * 8 + int(attrs["half"][p][0]) % 8)
            words.append(64 + (int(attrs["base"][p][0]) // 32) * 64 + (int(attrs["base"][p][1]) // 32) * 8
                         + int(attrs["base"][p][2]) // 32)
            words.append(600 + int(attrs["cen"][p][0]) * 4 % 512)
        attrs["words"] = [w % (vocab_size - 1) + 1 for w in words]
        shapes.append(attrs)
    items = []
    for s, attrs in enumerate(shapes):
        grid = make_voxel_grid_u8(rng, voxel_size, attrs) if voxel_size else None
        imgs = make_images_u8(rng, num_views, image_size, attrs) if num_views else None
        for c in range(captions_per_shape):
            tok = np.zeros(MAX_TOKENS, dtype=np.int32)
            w = list(attrs["words"])
            nnoise = int(rng.integers(0, 6))
            w += [int(x) for x in rng.integers(1, vocab_size, size=nnoise)]
            perm = rng.permutation(len(w))
            w = [w[i] for i in perm][:MAX_TOKENS]
            tok[:len(w)] = w
            items.append({"shape": s, "tokens": tok, "grid": grid, "imgs": imgs})
    return items


def collate_items(items, voxel: bool, views: bool) -> dict:
    d = {"model_id": [f"shape{it['shape']:05d}" for it in items], "category": ["synthetic"] * len(items),
         "tokens": torch.from_numpy(np.stack([it["tokens"] for it in items]))}
    if views:
        d["images"] = normalise_images(np.stack([it["imgs"] for it in items]))
    if voxel:
        locs, feats = [], []
        for i, it in enumerate(items):
            c, f = grid_to_sparse(it["grid"])
            locs.append(np.concatenate([np.full((c.shape[0], 1), i, dtype=np.int32), c], axis=1))
            feats.append(f)
        d["voxels"] = {"locs": torch.from_numpy(np.concatenate(locs)), "feats": torch.from_numpy(np.concatenate(feats))}
    return d


# ------------------------------------------------------------------------------------------------------
# Held-out retrieval set (SURVEY.md section 8d "Held-out retrieval set"): shapes are COMPOSITIONS of a few factors
# (body colour, body kind, body size, colour of a small second part); voxels, renderings and caption tokens are all
# functions of the factors plus per-instance noise, so a model trained on one draw of shapes retrieves UNSEEN shapes of
# the same factor space.  (A position factor was tried first and dropped: five conv + max-pool levels down to one site
# make the towers translation-invariant by construction - 84 % of the oracle's held-out errors were octant confusions.)
# ------------------------------------------------------------------------------------------------------
FACTOR_SIZES = (8, 2, 4, 8)                       # body colour, kind (box / ellipsoid), size class, cap colour -> 512 combinations
_PALETTE = ((220, 40, 40), (40, 200, 60), (50, 70, 230), (230, 210, 50), (200, 60, 210), (60, 210, 220), (240, 140, 40), (150, 150, 150))


def factor_words(f, vocab_size: int = DEFAULT_VOCAB):
    """One vocabulary word per factor value (disjoint ranges)."""
    c, k, s, c2 = f
    return [10 + c, 30 + k, 50 + s, 70 + c2]


def make_factor_shape(rng: np.random.Generator, f, V: int, num_views: int | None, S: int):
    """(RGBA u8 grid [4,V,V,V], u8 views [nv,3,S,S] | None) of one instance of factor combination f."""
    c, k, s, c2 = f
    half0 = max(2, V * (3 + 2 * s) // 32)                              # size class -> half extent (3, 5, 7, 9 at 32^3)
    half = np.array([half0 + int(rng.integers(0, 2)) for _ in range(3)])
    cen = np.array([V * 13 // 32, V // 2, V // 2]) + rng.integers(-1, 2, size=3)
    hcap = np.array([max(2, V // 16)] * 3)
    ccap = cen + np.array([half[0] + hcap[0] + 1, 0, 0])               # the cap sits on top of the body (never overlaps it)
    base, base2 = np.array(_PALETTE[c]), np.array(_PALETTE[c2])
    attrs = {"nparts": 2, "half": np.stack([half, hcap]), "cen": np.stack([cen, ccap]), "kind": np.array([k, 0]),
             "base": np.stack([base, base2])}
    grid = make_voxel_grid_u8(rng, V, attrs)
    imgs = None
    if num_views:
        img_attr = np.zeros((num_views, 3, 4), dtype=np.int64)
        for v in range(num_views):
            col = base if v % 2 == 0 else base2                         # even views show the body colour, odd views the cap colour
            for ch in range(3):
                img_attr[v, ch] = (1 + s, 1 + 2 * k, v, 40 + col[ch] * 140 // 255)
        imgs = make_images_u8(rng, num_views, S, {"img": img_attr})
    return grid, imgs


def make_factor_retrieval_set(num_shapes: int, captions_per_shape: int, voxel_size: int, num_views: int | None, image_size: int,
                              seed: int, distinct: bool, vocab_size: int = DEFAULT_VOCAB, max_noise_words: int = 4):
    """List of items {'shape', 'factors', 'tokens', 'grid', 'imgs'}.  distinct=True draws num_shapes DIFFERENT factor
    combinations (the held-out set: a caption then identifies exactly one shape); otherwise combinations repeat freely."""
    rng = np.random.default_rng(seed)
    ncomb = int(np.prod(FACTOR_SIZES))
    if distinct:
        assert num_shapes <= ncomb
        combos = rng.permutation(ncomb)[:num_shapes]
    else:
        combos = rng.integers(0, ncomb, size=num_shapes)
    items = []
    for sidx, cidx in enumerate(combos):
        f = np.unravel_index(int(cidx), FACTOR_SIZES)
        grid, imgs = make_factor_shape(rng, f, voxel_size, num_views, image_size)
        for _ in range(captions_per_shape):
            w = factor_words(f, vocab_size) + [int(x) for x in rng.integers(200, vocab_size, size=int(rng.integers(0, max_noise_words + 1)))]
            w = [w[i] for i in rng.permutation(len(w))]
            tok = np.zeros(MAX_TOKENS, dtype=np.int32)
            tok[:len(w)] = w
            items.append({"shape": sidx, "factors": tuple(int(x) for x in f), "tokens": tok, "grid": grid, "imgs": imgs})
    return items
