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
  * `TriCoLoNet.load_from_checkpoint` rebuilds the module from `hyper_parameters['cfg']` (tricolo_net.py:14) and reads it by
    ATTRIBUTE (`cfg.model.image_encoder`, tricolo_net.py:20-37) through `hydra.utils.instantiate`, i.e. it expects an OmegaConf
    DictConfig whose `_target_`s name the reference's classes.  save_reference_checkpoint therefore stores the net's config with
    every `tricolo_amd.*` target rewritten to its `tricolo.*` counterpart (FusedAdam -> torch.optim.Adam) as a DictConfig when
    omegaconf is importable (it is wherever the reference runs), else as a plain nested dict that `OmegaConf.create()` turns into
    one - in that case only the `state_dict` / `optimizer_states` halves are directly consumable by the reference;
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


_TARGET_MAP = (("tricolo_amd.optim.FusedAdam", "torch.optim.Adam"), ("tricolo_amd.", "tricolo."))


def reference_targets(node, back: bool = False):
    """A copy of a (nested) config dict with every `_target_` pointing at the reference's classes (tricolo.*, torch.optim.Adam) -
    or, with back=True, a reference config re-targeted at this package (what `load_reference_hparams` hands to TriCoLoNet)."""
    if isinstance(node, dict):
        out = {}
        for k, v in node.items():
            if k == "_target_" and isinstance(v, str):
                if back:
                    v = "tricolo_amd.optim.FusedAdam" if v == "torch.optim.Adam" else ("tricolo_amd." + v[len("tricolo."):] if v.startswith("tricolo.") else v)
                else:
                    for a, b in _TARGET_MAP:
                        if v.startswith(a):
                            v = b + v[len(a):]
                            break
                out[k] = v
            else:
                out[k] = reference_targets(v, back)
        return out
    if isinstance(node, (list, tuple)):
        return type(node)(reference_targets(v, back) for v in node)
    return node


def reference_hparams(cfg) -> dict:
    """hyper_parameters for a reference checkpoint from this package's config: {'cfg': DictConfig | dict} with reference targets."""
    plain = cfg.to_dict() if hasattr(cfg, "to_dict") else dict(cfg)
    plain = reference_targets(plain)
    try:
        from omegaconf import OmegaConf                       # present wherever the reference (hydra) is installed
        return {"cfg": OmegaConf.create(plain)}
    except ImportError:
        return {"cfg": plain}


def load_reference_hparams(ckpt):
    """The 'cfg' of a checkpoint's hyper_parameters as a tricolo_amd ConfigNode (targets mapped back to this package): what
    TriCoLoNet(cfg) of this package needs to rebuild the module the checkpoint was written from."""
    from .config import ConfigNode
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        ckpt = torch.load(ckpt, map_location="cpu", weights_only=False)
    cfg = ckpt["hyper_parameters"]["cfg"]
    try:
        from omegaconf import OmegaConf
        if OmegaConf.is_config(cfg):
            cfg = OmegaConf.to_container(cfg, resolve=True)
    except ImportError:
        pass
    return ConfigNode(reference_targets(dict(cfg), back=True))


def save_reference_checkpoint(net: torch.nn.Module, path, hyper_parameters=None, epoch: int = 0, global_step: int = 0,
                              optimizer=None, clip_state: dict | None = None, extra: dict | None = None) -> None:
    """Writes a Lightning-style file: the state dict under the reference's names (strictly loadable by the reference's modules),
    `optimizer_states` in torch.optim.Adam's format when `optimizer` is given (FusedAdam.state_dict() speaks it), `clip_state` as the
    `text_encoder.clip_model.*` tensors of a CLIP-text run, and the hyper-parameter slot of `save_hyperparameters()`
    (tricolo_net.py:14): by default reference_hparams(net._cfg) - reference `_target_`s, a DictConfig when omegaconf is importable
    (then `TriCoLoNet.load_from_checkpoint` of the reference can rebuild the module), a plain nested dict otherwise (module
    docstring)."""
    if hyper_parameters is None:
        cfg = getattr(net, "_cfg", None)
        if cfg is None:
            raise ValueError("save_reference_checkpoint: pass hyper_parameters={'cfg': ...} (the reference rebuilds the module from it)")
        hyper_parameters = reference_hparams(cfg)
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
