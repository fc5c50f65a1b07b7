"""Hydra-free composition / interpolation / instantiation for the TriCoLo config surface.

The reference drives everything through Hydra (/root/reference/train.py:17, config/config.yaml) and builds every
hot-path object with ``hydra.utils.instantiate`` on nodes carrying ``_target_`` (tricolo_net.py:26-40).  Hydra and
OmegaConf are not installed on the build or GPU boxes, so this module implements the subset the config tree uses:

  * a ``defaults`` list with ``_self_``, ``group: name`` entries and ``override hydra/...`` lines (ignored);
  * group files that themselves start with ``defaults: [base]`` (config/data/text2shape_*.yaml);
  * ``${a.b.c}`` interpolation against the root, and ``${hydra:runtime.cwd}``;
  * ``key=value`` / ``group=name`` command-line overrides (README.md:103-105 style);
  * ``instantiate(node, **kwargs)`` = import ``_target_`` and call it with the node's other keys plus kwargs.

When Hydra *is* importable the YAML files in ``tricolo_amd/config`` work with it unchanged.
"""
from __future__ import annotations

import importlib
import os
import re
from typing import Any

import yaml

_INTERP = re.compile(r"\$\{([^${}]+)\}")


class _Loader(yaml.SafeLoader):
    """YAML 1.2 style floats: PyYAML reads ``1e-6`` as a string, OmegaConf/Hydra as a float (config.yaml:53)."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?$|^[-+]?\.(?:inf|Inf|INF)$|^\.(?:nan|NaN|NAN)$"),
    list("-+0123456789."))


def _yaml_load(text):
    return yaml.load(text, Loader=_Loader)


class ConfigNode(dict):
    """dict with attribute access whose string leaves are interpolated against the root on read."""

    def __init__(self, data=None, root=None):
        super().__init__()
        object.__setattr__(self, "_root", root if root is not None else self)
        for k, v in (data or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, ConfigNode):
            v = ConfigNode(v, object.__getattribute__(self, "_root"))
        elif isinstance(v, ConfigNode):
            v._rebind(object.__getattribute__(self, "_root"))
        super().__setitem__(k, v)

    def _rebind(self, root):
        object.__setattr__(self, "_root", root)
        for v in dict.values(self):
            if isinstance(v, ConfigNode):
                v._rebind(root)

    def _resolve(self, v, depth=0):
        if not isinstance(v, str) or "${" not in v:
            if isinstance(v, list):
                return [self._resolve(x, depth) for x in v]
            return v
        if depth > 32:
            raise ValueError(f"interpolation cycle in {v!r}")
        root = object.__getattribute__(self, "_root")
        m = _INTERP.fullmatch(v)
        if m:                                         # whole-value interpolation keeps the type
            return self._resolve(_lookup(root, m.group(1)), depth + 1)
        return self._resolve(_INTERP.sub(lambda mm: str(self._resolve(_lookup(root, mm.group(1)), depth + 1)), v),
                             depth + 1)

    def __getitem__(self, k):
        return self._resolve(super().__getitem__(k))

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def to_dict(self) -> dict:
        return {k: (v.to_dict() if isinstance(v, ConfigNode) else v) for k, v in self.items()}


def _lookup(root: ConfigNode, path: str) -> Any:
    path = path.strip()
    if path.startswith("hydra:"):
        if path == "hydra:runtime.cwd":
            return os.getcwd()
        raise KeyError(f"unsupported resolver {path}")
    node: Any = root
    for part in path.split("."):
        node = dict.__getitem__(node, part) if isinstance(node, ConfigNode) else node[part]
    return node


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _load_group_file(config_dir: str, group: str, name: str) -> dict:
    with open(os.path.join(config_dir, group, f"{name}.yaml")) as f:
        raw = _yaml_load(f.read()) or {}
    out: dict = {}
    for d in raw.pop("defaults", []) or []:
        if isinstance(d, str) and d != "_self_":
            _merge(out, _load_group_file(config_dir, group, d))
    return _merge(out, raw)


def default_config_dir() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")


def compose(config_dir: str | None = None, config_name: str = "config", overrides=()) -> ConfigNode:
    """Compose ``<config_dir>/<config_name>.yaml`` with its defaults list and ``overrides``."""
    config_dir = config_dir or default_config_dir()
    with open(os.path.join(config_dir, f"{config_name}.yaml")) as f:
        raw = _yaml_load(f.read()) or {}
    defaults = raw.pop("defaults", []) or []
    raw.pop("hydra", None)
    group_choice: dict[str, str] = {}
    order: list[str] = []
    for d in defaults:
        if d == "_self_":
            order.append("_self_")
        elif isinstance(d, dict):
            for g, n in d.items():
                if str(g).startswith("override "):
                    continue
                group_choice[g] = n
                order.append(g)
    plain = []
    for ov in overrides:
        k, _, v = ov.partition("=")
        k = k.lstrip("+")
        if k in group_choice or os.path.isdir(os.path.join(config_dir, k)):
            group_choice[k] = v
            if k not in order:
                order.append(k)
        else:
            plain.append((k, _yaml_load(v) if v != "" else None))
    if "_self_" not in order:
        order.append("_self_")
    cfg: dict = {}
    for item in order:
        if item == "_self_":
            _merge(cfg, raw)
        else:
            _merge(cfg, {item: _load_group_file(config_dir, item, group_choice[item])})
    for k, v in plain:
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return ConfigNode(cfg)


def locate(path: str):
    mod, _, attr = path.rpartition(".")
    return getattr(importlib.import_module(mod), attr)


def instantiate(node, *args, **kwargs):
    """``hydra.utils.instantiate`` for ``_target_`` nodes (recursive for nested ``_target_`` dicts)."""
    if node is None:
        return None
    if "_target_" not in node:
        raise ValueError("instantiate() needs a node with _target_")
    target = locate(node["_target_"])
    kw = {}
    for k, v in node.items():
        if k in ("_target_", "_partial_", "_recursive_", "_convert_"):
            continue
        kw[k] = instantiate(v) if isinstance(v, dict) and "_target_" in v else v
    kw.update(kwargs)
    return target(*args, **kw)
