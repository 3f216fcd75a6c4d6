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
    def load_from_checkpoint(cls, path, map_location="cpu", **overrides):
        """Lightning checkpoint format: {'state_dict', 'hyper_parameters', ...} (download.py:56-61)."""
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(overrides)
        model = cls(**hp)
        model.load_state_dict(ckpt["state_dict"], strict=False)
        return model
