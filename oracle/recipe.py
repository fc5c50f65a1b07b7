"""Deterministic weight recipe keyed by state-dict name.  TEST INFRASTRUCTURE.

ImageNet / published checkpoints cannot be downloaded here, and module constructors of the reference, the oracle
and the product draw from torch's RNG in different orders.  Parity tests therefore fill every state dict from this
recipe: the value of a tensor depends only on (seed, key name, shape), so loading it into the reference class, the
oracle restatement and the HIP-backed module gives bit-identical weights - and exercises the state-dict key
compatibility promised in SURVEY.md section 8(b).
"""
import zlib

import numpy as np
import torch


def recipe_tensor(key: str, shape, dtype=torch.float32, seed: int = 123) -> torch.Tensor:
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if key.endswith("running_mean"):
        return torch.zeros(shape, dtype=dtype)
    if key.endswith("running_var"):
        return torch.ones(shape, dtype=dtype)
    rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
    u = rng.integers(-(1 << 20), (1 << 20) + 1, size=shape).astype(np.float64) / float(1 << 20)   # U[-1,1], exact
    if len(shape) == 1:
        if key.endswith("weight"):                      # BatchNorm gamma
            val = 1.0 + 0.1 * u
        else:                                           # any bias (BN beta, Linear, GRU)
            val = 0.05 * u
    else:
        fan_in = int(np.prod(shape[1:]))
        val = u * np.sqrt(6.0 / fan_in)
        if key.endswith("embedding_layer.weight"):
            val = u * 0.5
            val[0] = 0.0                                # padding_idx=0 row (bigru.py:10)
    return torch.from_numpy(val.astype(np.float32)).to(dtype)


def fill_module(module: torch.nn.Module, seed: int = 123, prefix: str = "") -> None:
    """Overwrite every parameter / buffer of ``module`` in place from the recipe (keys = state_dict names)."""
    sd = module.state_dict()
    new = {k: recipe_tensor(prefix + k, v.shape, v.dtype if v.dtype.is_floating_point else torch.float32, seed)
           if v.dtype.is_floating_point else recipe_tensor(prefix + k, v.shape, seed=seed) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)


def sample_indices(numel: int, k: int = 16) -> np.ndarray:
    """Deterministic probe positions inside a flattened tensor (used for compact gradient / weight fixtures)."""
    if numel <= k:
        return np.arange(numel)
    return np.unique(np.round(np.linspace(0, numel - 1, k)).astype(np.int64))


def probe(t: torch.Tensor, k: int = 16):
    """(L2 norm as float64, sampled entries as float32) of a tensor."""
    flat = t.detach().reshape(-1).to(torch.float64)
    idx = sample_indices(flat.numel(), k)
    return float(flat.norm().item()), flat[torch.from_numpy(idx)].to(torch.float32).numpy()
