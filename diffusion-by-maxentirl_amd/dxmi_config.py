"""Minimal stand-in for the OmegaConf + hydra.utils.instantiate pair the reference scripts use
(train_cifar10.py:228-259, generate_cifar10.py:119-151): YAML load, recursive merge, attribute
access, `_target_` instantiation.  Neither omegaconf nor hydra is installed in the target image,
and the reference's YAML files must be accepted unchanged."""
import importlib

import yaml


class Cfg(dict):
    """dict with attribute access and .get; nested dicts are wrapped on read."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if isinstance(v, dict) and not isinstance(v, Cfg):
            v = Cfg(v)
            dict.__setitem__(self, k, v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default


def load(path):
    with open(path) as f:
        return Cfg(yaml.safe_load(f) or {})


def merge(base, override):
    """Recursive dict merge (OmegaConf.merge semantics for plain mappings); returns a new Cfg."""
    out = Cfg(dict(base))
    for k, v in override.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = merge(out[k], v)
        else:
            out[k] = v
    return out


def to_container(cfg):
    if isinstance(cfg, dict):
        return {k: to_container(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [to_container(v) for v in cfg]
    return cfg


def save(cfg, path):
    with open(path, "w") as f:
        yaml.safe_dump(to_container(cfg), f, sort_keys=False)


def _locate(dotted):
    mod, _, name = dotted.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node, **kwargs):
    """hydra.utils.instantiate for the subset the configs use: a mapping with `_target_` becomes a
    call of that dotted callable; nested `_target_` mappings are instantiated first (recursive);
    keyword overrides win; None stays None."""
    if node is None:
        return None
    if not isinstance(node, dict) or "_target_" not in node:
        raise ValueError(f"instantiate: node has no _target_: {node!r}")
    args = {}
    for k, v in node.items():
        if k == "_target_":
            continue
        args[k] = instantiate(v) if isinstance(v, dict) and "_target_" in v else to_container(v)
    args.update(kwargs)
    return _locate(node["_target_"])(**args)
