"""Checkpoint compatibility with the reference.

The reference's checkpoints are PyTorch-Lightning files (`trainer.fit` + `ModelCheckpoint`, train.py:44-52) read back by
`TriCoLoNet.load_from_checkpoint(ckpt_path)` (test.py:19-29, README.md:124-129): a pickle with a `state_dict` whose keys
are `text_encoder.*`, `image_encoder.*`, `voxel_encoder.*`.  The modules of this package keep exactly those names and
layouts (spconv `[Cout,kd,kh,kw,Cin]`, torchvision `[Cout,Cin,kh,kw]`, `nn.GRU` / `nn.Linear` / `nn.BatchNorm*`), so
loading is `load_state_dict(strict=True)`; the only deliberate difference is `voxel_encoder.mlp.0.weight` at 32^3
(in_features 512 instead of the reference's hard-coded 4096) - reported, not hidden.

Known differences, handled explicitly:
  * the reference's CLIPTextEncoder registers the frozen CLIP model as a submodule (clip_text.py:8), so its checkpoints carry
    `text_encoder.clip_model.*` keys the MLP never uses: they are dropped on load (`ignored` in the return value) and, when
    a checkpoint is written for the reference, copied through from `clip_state` if the caller has them;
  * `TriCoLoNet.load_from_checkpoint` rebuilds the module from `hyper_parameters['cfg']` (tricolo_net.py:14), so
    save_reference_checkpoint stores the net's own config there;
  * CLIPImageEncoder and TripletLoss (config.yaml:83-96) are outside the hot path and are not built: a config that names
    them raises in tricolo_amd.config.instantiate (no such class under tricolo_amd), it does not load silently.
Pure host code: no kernel is involved.
"""
from __future__ import annotations

import torch


def extract_state_dict(obj) -> dict:
    """Lightning checkpoint dict, bare state dict, or a path to either -> {name: tensor}."""
    if isinstance(obj, (str, bytes)) or hasattr(obj, "__fspath__"):
        obj = torch.load(obj, map_location="cpu", weights_only=False)
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    if not isinstance(obj, dict):
        raise TypeError("expected a Lightning checkpoint, a state dict or a path to one")
    return obj


def load_reference_checkpoint(net: torch.nn.Module, ckpt, strict: bool = True):
    """Loads a reference checkpoint into a tricolo_amd TriCoLoNet.  Returns (missing_keys, unexpected_keys, mismatched)
    where mismatched = [(name, ckpt shape, module shape)]; raises when strict and anything is left over."""
    sd = extract_state_dict(ckpt)
    sd = {k: v for k, v in sd.items() if ".clip_model." not in k}      # frozen CLIP weights of clip_text.py:8 (unused by the MLP)
    own = net.state_dict()
    mismatched = [(k, tuple(v.shape), tuple(own[k].shape)) for k, v in sd.items() if k in own and tuple(v.shape) != tuple(own[k].shape)]
    if mismatched and strict:
        raise RuntimeError("checkpoint tensors with a different shape: " + ", ".join(f"{k} {a} vs {b}" for k, a, b in mismatched))
    skip = {k for k, _, _ in mismatched}
    res = net.load_state_dict({k: v for k, v in sd.items() if k not in skip}, strict=False)
    missing = [k for k in res.missing_keys if k not in skip]
    if strict and (missing or res.unexpected_keys):
        raise RuntimeError(f"checkpoint does not match the module: missing {missing}, unexpected {list(res.unexpected_keys)}")
    return missing, list(res.unexpected_keys), mismatched


def save_reference_checkpoint(net: torch.nn.Module, path, hyper_parameters=None, epoch: int = 0, global_step: int = 0,
                              optimizer=None, clip_state: dict | None = None, extra: dict | None = None) -> None:
    """Writes the minimal Lightning-style file `TriCoLoNet.load_from_checkpoint` of the reference reads: the state dict
    under the reference's names plus the hyper-parameter slot (`save_hyperparameters()`, tricolo_net.py:14; default: the
    net's own cfg as a plain dict under 'cfg').  `optimizer` adds Lightning's `optimizer_states` list (torch.optim.Adam
    format - FusedAdam.state_dict() speaks it), `clip_state` the `text_encoder.clip_model.*` tensors of a CLIP-text run."""
    if hyper_parameters is None:
        cfg = getattr(net, "_cfg", None)
        if cfg is None:
            raise ValueError("save_reference_checkpoint: pass hyper_parameters={'cfg': ...} (the reference rebuilds the module from it)")
        hyper_parameters = {"cfg": cfg.to_dict() if hasattr(cfg, "to_dict") else cfg}
    if "cfg" not in hyper_parameters:
        raise ValueError("hyper_parameters must hold 'cfg' (TriCoLoNet(**hyper_parameters), tricolo_net.py:12-14)")
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    if clip_state:
        sd.update({(k if k.startswith("text_encoder.clip_model.") else "text_encoder.clip_model." + k): v.detach().cpu()
                   for k, v in clip_state.items()})
    doc = {"state_dict": sd, "epoch": epoch, "global_step": global_step, "hyper_parameters": hyper_parameters,
           "pytorch-lightning_version": "2.0.0"}
    if optimizer is not None:
        doc["optimizer_states"] = [optimizer.state_dict()]
    if extra:
        doc.update(extra)
    torch.save(doc, path)
