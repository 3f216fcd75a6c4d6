"""Minimal config objects: the reference drives everything through OmegaConf (absent here), and its
constructor needs attribute *and* item access, `in`, `.get`, `hasattr` and in-place defaults
(mebt/transformer.py:83-135, utils.py:3-7).  `load_config` accepts the shipped YAMLs unchanged
(configs/{stl,taichi,ucf}/mebt_{16f,128f}.yaml) plus `a.b.c=value` overrides
(train_transformer.py:25-27)."""
import copy

import yaml


class AttrDict(dict):
    """dict with attribute access; nested dicts are converted on the way in."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, list):
            return [AttrDict._wrap(i) for i in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def merge(base, other):
    for k, v in other.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            merge(base[k], v)
        else:
            base[k] = v
    return base


def _parse_scalar(text):
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def load_config(paths=(), dotlist=()):
    cfg = AttrDict()
    for p in paths:
        with open(p) as f:
            merge(cfg, AttrDict(yaml.safe_load(f) or {}))
    for item in dotlist:
        key, _, val = item.partition("=")
        node = cfg
        parts = key.lstrip("-").split(".")
        for part in parts[:-1]:
            if part not in node or not isinstance(node[part], dict):
                node[part] = AttrDict()
            node = node[part]
        node[parts[-1]] = _parse_scalar(val)
    return cfg


def instantiate_from_config(config):
    """Counterpart of the reference's top-level utils.instantiate_from_config (utils.py:3-7): builds
    `config['target']` (legacy 'tats.' prefix rewritten to 'mebt.') with `config['params']`."""
    import importlib
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    target = config["target"].replace("tats.", "mebt.")
    config["target"] = target
    module, cls = target.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)(**config.get("params", dict()))
