"""pytorch_lightning is not installed on the target image; the reference's Net2NetTransformer is a
pl.LightningModule (mebt/transformer.py:60).  This base class provides the handful of members the
hot path touches (save_hyperparameters :146, log :679,736-745, global_step :231,244,
current_epoch :334, device :341, trainer.global_step/max_steps :666-672) so that the launcher in
mebt_amd/train.py — and a real Lightning Trainer, if one is installed — can drive the module."""
import pickle
import types

import torch
import torch.nn as nn


# ---- reading the reference's real checkpoints -----------------------------------------------------------------------------
# train_transformer.py:27-33 builds the model from OmegaConf nodes and `save_hyperparameters()` (transformer.py:146) pickles them
# into the checkpoint: `hyper_parameters` = {'transformer_config': DictConfig, 'first_stage_config': DictConfig, 'mask_config':
# DictConfig, ...}, possibly wrapped in pytorch_lightning's AttributeDict.  Neither package is installed on the target image, and
# a checkpoint must not need them: the unpickler below stands in for every class of `omegaconf.*` / `pytorch_lightning.*` /
# `lightning*` with an inert record of the pickled state, and `plain()` rebuilds ordinary containers from OmegaConf's layout
# (a container keeps its children in `_content`: a dict / list of nodes; a value node keeps its value in `_val`).  Everything
# else a checkpoint may name is on an exact (module, attribute) allow-list (tensor / storage reconstruction, numpy array
# reconstruction, OrderedDict, typing markers, argparse.Namespace, plain builtin types); any other global — and any dotted
# attribute path — is refused instead of imported.
class _Record:
    """inert stand-in for an instance of a class that is not installed: keeps the pickled state, runs no code of that class"""

    def __new__(cls, *args, **kwargs):
        o = object.__new__(cls)
        o.__dict__["_args"] = args
        return o

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):    # (dict state, slots state)
            state = {**(state[0] or {}), **state[1]}
        self.__dict__["_state"] = state

    def __call__(self, *args, **kwargs):           # enum-style `Class(value)` reconstruction
        return self


class _RecordDict(dict):
    """the same for dict subclasses (pytorch_lightning.utilities.parsing.AttributeDict, omegaconf's OrderedDict-like helpers)"""

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.update(state)


_STANDIN_PREFIXES = ("omegaconf", "pytorch_lightning", "lightning", "lightning_fabric")
_DICT_LIKE = {"AttributeDict"}

# Exact (module, name) allow-list (a root-module test is no boundary — pickle protocol 4 resolves dotted names, so
# ('torch', 'os.system') went through it).  A name with a '.' in it is always refused.
_TORCH_STORAGES = {"FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage", "ShortStorage",
                   "CharStorage", "ByteStorage", "BoolStorage", "ComplexFloatStorage", "ComplexDoubleStorage", "UntypedStorage"}
_ALLOWED = {
    "torch._utils": {"_rebuild_tensor_v2", "_rebuild_tensor", "_rebuild_parameter", "_rebuild_parameter_with_state", "_rebuild_device_tensor_from_numpy"},
    "torch": _TORCH_STORAGES | {"Size", "device", "Tensor", "dtype"},            # + every torch.dtype singleton, see find_class
    "torch.nn.parameter": {"Parameter"},
    "collections": {"OrderedDict"},
    "numpy": {"ndarray", "dtype"},
    "numpy.core.multiarray": {"_reconstruct", "scalar"}, "numpy._core.multiarray": {"_reconstruct", "scalar"},
    "numpy.core.numeric": {"_frombuffer"}, "numpy._core.numeric": {"_frombuffer"},
    "argparse": {"Namespace"},
    "typing": {"Any", "Union", "Optional", "Dict", "List", "Tuple"},
    "builtins": {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray", "complex", "slice", "range", "object"},
    "copyreg": {"_reconstructor"},          # object.__new__(cls) of an allow-listed / stand-in class (protocol-2 instances)
    "_codecs": {"encode"},                  # protocol-2 spelling of bytes
    "pathlib": {"PosixPath", "PurePosixPath", "Path"},
    "mebt_amd.config": {"AttrDict"},        # hyper-parameters of checkpoints this package saved itself
}
_ALLOWED["__builtin__"] = _ALLOWED["builtins"]          # protocol-2 spellings (torch.save default); pickle maps them to the py3 modules itself
_ALLOWED["copy_reg"] = _ALLOWED["copyreg"]


def _standin_class(module, name):
    base = _RecordDict if name in _DICT_LIKE else _Record
    return type(name, (base,), {"__module__": module, "_standin_for": f"{module}.{name}"})


class TolerantUnpickler(pickle.Unpickler):
    """pickle.Unpickler that resolves ONLY the globals of `_ALLOWED` (exact module and attribute name) and turns every class of
    omegaconf / pytorch_lightning into an inert record; anything else raises UnpicklingError before it is imported."""

    def find_class(self, module, name):
        if "." in name:
            raise pickle.UnpicklingError(f"checkpoint names {module}.{name} (a dotted attribute path): refused")
        root = module.split(".", 1)[0]
        if root in _STANDIN_PREFIXES:
            return _standin_class(module, name)
        ok = name in _ALLOWED.get(module, ())
        if not ok and module == "torch":        # dtype singletons pickle as the global torch.<name>
            ok = isinstance(getattr(torch, name, None), torch.dtype)
        if ok:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"checkpoint names {module}.{name}, which is outside what a MeBT checkpoint may contain "
                                     "(tensor / storage reconstruction, numpy arrays, stdlib containers; omegaconf and pytorch_lightning "
                                     "classes are read through stand-ins)")


_tolerant_pickle = types.SimpleNamespace(__name__="pickle", Unpickler=TolerantUnpickler, load=lambda f, **kw: TolerantUnpickler(f, **kw).load(),
                                         loads=pickle.loads, dump=pickle.dump, dumps=pickle.dumps, HIGHEST_PROTOCOL=pickle.HIGHEST_PROTOCOL,
                                         UnpicklingError=pickle.UnpicklingError, PicklingError=pickle.PicklingError)


def plain(v, _seen=None):
    """OmegaConf / Lightning stand-ins (and anything dict- or list-like) -> plain dict / list / scalar, recursively"""
    if isinstance(v, _Record):
        st = v.__dict__.get("_state") or {}
        if isinstance(st, dict) and "_content" in st:              # DictConfig / ListConfig
            c = st["_content"]
            if isinstance(c, dict):
                return {plain(k): plain(x) for k, x in c.items()}
            if isinstance(c, (list, tuple)):
                return [plain(x) for x in c]
            return plain(c)                                        # a None / missing / interpolation string container
        if isinstance(st, dict) and "_val" in st:                  # AnyNode / StringNode / IntegerNode / FloatNode / BooleanNode / EnumNode
            return plain(st["_val"])
        a = v.__dict__.get("_args") or ()
        return plain(a[0]) if len(a) == 1 else None                # an enum member or an opaque helper: its value, or nothing
    if isinstance(v, dict):
        return {plain(k): plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)) and not isinstance(v, str):
        return [plain(x) for x in v]
    if hasattr(v, "__dict__") and type(v).__name__ == "Namespace":      # argparse.Namespace hyper-parameters
        return {k: plain(x) for k, x in vars(v).items()}
    return v


def load_checkpoint_file(path, map_location="cpu"):
    """torch.load of a Lightning checkpoint that may pickle OmegaConf / Lightning objects, without those packages"""
    ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_tolerant_pickle)
    if "hyper_parameters" in ckpt:
        ckpt["hyper_parameters"] = plain(ckpt["hyper_parameters"])
    return ckpt


class _TrainerState:
    def __init__(self):
        self.global_step = 0
        self.max_steps = 0


class LightningModuleShim(nn.Module):
    def __init__(self):
        super().__init__()
        self.global_step = 0
        self.current_epoch = 0
        self.trainer = _TrainerState()
        self.logged = {}
        self.hparams = {}
        self.logger = None                 # anything with `.experiment.add_video(tag, vid, step, fps=)` / `.experiment.flush()` (VideoLogger below)

    def save_hyperparameters(self, *args, **kwargs):
        self.hparams.update(kwargs)

    def log(self, name, value, **kwargs):
        self.logged[name] = value

    # Lightning's loop hooks the reference overrides (transformer.py:332-351); the launcher calls them at the same places
    def on_train_epoch_start(self):
        pass

    def on_validation_epoch_start(self):
        pass

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu", strict=False, **overrides):
        """Lightning checkpoint format (download.py:56-61): {'state_dict', 'hyper_parameters', 'global_step', 'epoch',
        'pytorch-lightning_version', ...}.  `hyper_parameters` holds the constructor arguments `save_hyperparameters()`
        recorded (transformer.py:146): config nodes arrive as plain nested dicts OR as the OmegaConf DictConfig / ListConfig
        objects the reference's launcher pickles (read without omegaconf / pytorch_lightning installed: load_checkpoint_file)
        and are wrapped into attribute-access configs; Lightning's bookkeeping keys are ignored."""
        from .config import AttrDict
        ckpt = load_checkpoint_file(path, map_location=map_location)

        def wrap(v):
            if isinstance(v, AttrDict):
                return v
            if isinstance(v, dict) or (hasattr(v, "keys") and hasattr(v, "__getitem__")):
                return AttrDict({k: wrap(v[k]) for k in v.keys()})
            if isinstance(v, (list, tuple)) and not isinstance(v, str):
                return [wrap(i) for i in v]
            return v

        import inspect
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        hp = {k: wrap(v) for k, v in dict(ckpt.get("hyper_parameters", {})).items() if k in accepted}
        hp.update(overrides)
        hp.pop("ckpt_path", None)                  # the weights come from THIS file, not from a path recorded at training time
        model = cls(**hp)
        model.load_state_dict(ckpt["state_dict"], strict=strict)
        model.global_step = int(ckpt.get("global_step", 0))
        model.current_epoch = int(ckpt.get("epoch", 0))
        model.trainer.global_step = model.global_step
        return model


class VideoLogger:
    """The slice of Lightning's TensorBoardLogger the reference module touches (`self.logger.experiment.add_video(...)`,
    `.flush()`, transformer.py:349-350) without TensorBoard: videos `[N, T, C, H, W]` in [0, 1] are written as uint8 `.npy` files
    `<dir>/<tag>_<step>.npy`, scalars appended to `<dir>/scalars.tsv`."""

    def __init__(self, log_dir):
        import os
        self.log_dir = log_dir
        os.makedirs(log_dir, exist_ok=True)
        self.experiment = self
        self.videos = []

    def add_video(self, tag, vid_tensor, global_step=None, fps=4):
        import os
        import numpy as np
        arr = (vid_tensor.detach().float().clamp(0, 1) * 255.0).round().to(torch.uint8).cpu().numpy()
        path = os.path.join(self.log_dir, f"{tag.replace('/', '_')}_{int(global_step or 0):06d}.npy")
        np.save(path, arr)
        self.videos.append((tag, path, int(global_step or 0), fps))

    def add_scalar(self, tag, value, global_step=None):
        import os
        with open(os.path.join(self.log_dir, "scalars.tsv"), "a") as f:
            f.write(f"{tag}\t{int(global_step or 0)}\t{float(value):.8g}\n")

    def flush(self):
        pass

