"""pytorch_lightning is not installed on the target image; the reference's Net2NetTransformer is a
pl.LightningModule (mebt/transformer.py:60).  This base class provides the handful of members the
hot path touches (save_hyperparameters :146, log :679,736-745, global_step :231,244,
current_epoch :334, device :341, trainer.global_step/max_steps :666-672) so that the launcher in
mebt_amd/train.py — and a real Lightning Trainer, if one is installed — can drive the module."""
import torch
import torch.nn as nn


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

    def save_hyperparameters(self, *args, **kwargs):
        self.hparams.update(kwargs)

    def log(self, name, value, **kwargs):
        self.logged[name] = value

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu", strict=False, **overrides):
        """Lightning checkpoint format (download.py:56-61): {'state_dict', 'hyper_parameters', 'global_step', 'epoch',
        'pytorch-lightning_version', ...}.  `hyper_parameters` holds the constructor arguments `save_hyperparameters()`
        recorded (transformer.py:146): config nodes arrive as plain nested dicts (or anything dict-like) and are wrapped
        into attribute-access configs; Lightning's bookkeeping keys are ignored.  A checkpoint that pickles OmegaConf
        objects needs `omegaconf` importable to be unpickled at all — that failure is reported as such."""
        from .config import AttrDict
        try:
            ckpt = torch.load(path, map_location=map_location, weights_only=False)
        except ModuleNotFoundError as e:
            raise RuntimeError(f"{path} pickles objects of the module '{e.name}', which is not installed here: install it, or "
                               "re-save the checkpoint with `hyper_parameters` as plain nested dicts") from e

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
