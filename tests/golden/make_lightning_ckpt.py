"""A checkpoint laid out the way the reference's training run writes it (train_transformer.py:27-33 -> transformer.py:146
`save_hyperparameters()` -> Lightning's ModelCheckpoint): `hyper_parameters` holds OmegaConf DictConfig / ListConfig objects,
optionally inside pytorch_lightning's AttributeDict.  Neither package exists on this image, so this script registers stand-in
classes UNDER THE REAL MODULE AND CLASS NAMES with OmegaConf's pickle layout — a container's children in `_content` (dict /
list of nodes), a value node's value in `_val`, `_metadata` dataclass records, `_parent` back-references (cycles) — and pickles
them with torch.save; the stand-in modules are removed again before the loader under test runs.  Used by
tests/test_host_cpu.py::test_load_reference_checkpoint_with_omegaconf_hparams; the checkpoint itself is written to a temp
directory (8 MB of closed-form weights), only this recipe is committed."""
import sys
import types
import typing


def _install():
    made = {}

    def module(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        made[name] = m
        return m

    root, base, dc, lc, nodes = module("omegaconf"), module("omegaconf.base"), module("omegaconf.dictconfig"), module("omegaconf.listconfig"), module("omegaconf.nodes")
    pl, plu, plp = module("pytorch_lightning"), module("pytorch_lightning.utilities"), module("pytorch_lightning.utilities.parsing")

    def cls(mod, name, bases=(object,), body=None):
        c = type(name, bases, dict(body or {}, __module__=mod.__name__))
        setattr(mod, name, c)
        return c

    Metadata = cls(base, "Metadata")
    ContainerMetadata = cls(base, "ContainerMetadata", (Metadata,))

    def getstate(self):                         # BaseContainer.__getstate__ / Node pickling: the instance dict minus the flags cache
        d = dict(self.__dict__)
        d.pop("_flags_cache", None)
        return d

    Node = cls(base, "Node", body={"__getstate__": getstate, "__setstate__": lambda self, d: self.__dict__.update(d)})
    DictConfig = cls(dc, "DictConfig", (Node,))
    ListConfig = cls(lc, "ListConfig", (Node,))
    kinds = {str: cls(nodes, "StringNode", (Node,)), int: cls(nodes, "IntegerNode", (Node,)), float: cls(nodes, "FloatNode", (Node,)),
             bool: cls(nodes, "BooleanNode", (Node,))}
    AnyNode = cls(nodes, "AnyNode", (Node,))
    AttributeDict = cls(plp, "AttributeDict", (dict,))
    root.DictConfig, root.ListConfig = DictConfig, ListConfig

    def meta(container, key):
        m = (ContainerMetadata if container else Metadata)()
        m.__dict__.update(ref_type=typing.Any, object_type=dict if container else None, optional=True, key=key, flags={}, flags_root=False,
                          resolver_cache={})
        if container:
            m.__dict__.update(key_type=typing.Any, element_type=typing.Any)
        return m

    def wrap(v, parent=None, key=None, typed=False):
        if isinstance(v, dict):
            n = DictConfig()
            n.__dict__.update(_metadata=meta(True, key), _parent=parent, _flags_cache={}, _content={})
            n.__dict__["_content"] = {k: wrap(x, n, k, typed) for k, x in v.items()}
            return n
        if isinstance(v, (list, tuple)):
            n = ListConfig()
            n.__dict__.update(_metadata=meta(True, key), _parent=parent, _flags_cache={}, _content=[])
            n.__dict__["_content"] = [wrap(x, n, i, typed) for i, x in enumerate(v)]
            return n
        n = (kinds.get(type(v), AnyNode) if typed else AnyNode)()
        n.__dict__.update(_metadata=meta(False, key), _parent=parent, _flags_cache=None, _val=v)
        return n

    return made, wrap, AttributeDict


def write(path, state_dict, hparams, flavour="plain"):
    """flavour 'plain': hyper_parameters = dict of DictConfig (untyped AnyNode leaves); 'attributedict': Lightning's AttributeDict
    around DictConfigs with typed leaf nodes"""
    import torch
    saved = {k: sys.modules.get(k) for k in ("omegaconf", "omegaconf.base", "omegaconf.dictconfig", "omegaconf.listconfig", "omegaconf.nodes",
                                              "pytorch_lightning", "pytorch_lightning.utilities", "pytorch_lightning.utilities.parsing")}
    made, wrap, AttributeDict = _install()
    try:
        hp = {k: (wrap(v, typed=flavour != "plain") if isinstance(v, (dict, list)) else v) for k, v in hparams.items()}
        if flavour != "plain":
            hp = AttributeDict(hp)
        torch.save({"epoch": 11, "global_step": 50000, "pytorch-lightning_version": "1.5.4", "state_dict": state_dict, "hyper_parameters": hp,
                    "hparams_name": "kwargs", "optimizer_states": [], "lr_schedulers": [], "callbacks": {}, "loops": {}}, path)
    finally:
        for k in made:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
    return path
