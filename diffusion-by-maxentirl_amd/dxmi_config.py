"""Minimal stand-in for the OmegaConf + hydra.utils.instantiate pair the reference scripts use
(train_cifar10.py:228-259, generate_cifar10.py:119-151): YAML load, recursive merge, attribute
access, `_target_` instantiation.  Neither omegaconf nor hydra is installed in the target image,
and the reference's YAML files must be accepted unchanged
(tests/test_cli_config.py loads all ten of them when the reference checkout is present)."""
import importlib
import re

import yaml


class _Loader(yaml.SafeLoader):
    """PyYAML's SafeLoader with OmegaConf's scalar rules (omegaconf/_utils.py get_yaml_loader, which
    train_cifar10.py:228-233 goes through): floats by the YAML-1.2 pattern, so `1e-7` / `1e-5` (no dot, unsigned
    exponent) are numbers and not the strings YAML 1.1 makes of them; no implicit timestamps; duplicate keys are
    an error.  `Null`, `True`, `yes` ... resolve as in PyYAML (OmegaConf keeps those rules)."""


_Loader.yaml_implicit_resolvers = {
    k: [(tag, rx) for tag, rx in v if tag != "tag:yaml.org,2002:timestamp"]
    for k, v in yaml.SafeLoader.yaml_implicit_resolvers.items()}
_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"""^(?:
         [-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
        |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
        |\.[0-9_]+(?:[eE][-+][0-9]+)?
        |[-+]?[0-9][0-9_]*(?::[0-5]?[0-9])+\.[0-9_]*
        |[-+]?\.(?:inf|Inf|INF)
        |\.(?:nan|NaN|NAN))$""", re.X),
    list("-+0123456789."))


def _mapping_no_duplicates(loader, node, deep=False):
    seen = set()
    for key_node, _ in node.value:
        # as OmegaConf's loader: only plain keys are compared (a merge key `<<: *anchor` has tag:yaml.org,2002:merge, for which
        # SafeLoader has no constructor — construct_mapping below flattens it)
        if key_node.tag == "tag:yaml.org,2002:merge" or not isinstance(key_node, yaml.ScalarNode):
            continue
        key = loader.construct_object(key_node, deep=deep)
        if key in seen:
            raise yaml.constructor.ConstructorError(
                "while constructing a mapping", node.start_mark, f"found duplicate key {key!r}", key_node.start_mark)
        seen.add(key)
    return yaml.SafeLoader.construct_mapping(loader, node, deep)


_Loader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG,
                        lambda loader, node: _mapping_no_duplicates(loader, node, deep=True))


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
        return Cfg(yaml.load(f, Loader=_Loader) or {})


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
