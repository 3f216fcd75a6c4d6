"""Shared helpers of the GPU parity tests (test infrastructure; may import oracle/)."""
import os
import numpy as np
import torch

from oracle import mebt_oracle as orc

G = os.path.join(os.path.dirname(__file__), "golden")
_params = {}


def load_golden(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def params_for(name):
    from tests.golden import make_golden as mg
    if name not in _params:
        _params[name] = orc.closed_form_params(mg.oracle_cfg(name))
    return _params[name]


def build_native(cfg, dtype, P=None, device="cuda"):
    """NativeModel (HIP engine) holding the closed-form weights of `cfg`."""
    from mebt_amd.engine import NativeModel
    nm = NativeModel(cfg.n_layer, cfg.n_head, cfg.n_embd, cfg.vocab_size, cfg.sos_emb, cfg.block_size, cfg.mode,
                     dtype=dtype, label_smoothing=cfg.label_smoothing)
    nm.allocate(device)
    if P is None:
        P = orc.closed_form_params(cfg)
    views = nm.views(orc.param_shapes(cfg))
    with torch.no_grad():
        for k, v in P.items():
            views[k].copy_(v.detach())
    nm.sync_lowp(force=True)
    return nm
