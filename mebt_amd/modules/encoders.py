"""Conditioning-stage shims (reference mebt/modules/encoders.py).  Only SOSProvider is reachable:
the model is unconditional and `init_cond_stage_from_ckpt` uses it to set cond_stage_vocab_size = 0
(mebt/transformer.py:204-212)."""
import torch
import torch.nn as nn


class SOSProvider(nn.Module):
    def __init__(self, sos_token, quantize_interface=True):
        super().__init__()
        self.sos_token = sos_token
        self.quantize_interface = quantize_interface

    def encode(self, x, **kwargs):
        c = torch.full((x.shape[0], 1), self.sos_token, dtype=torch.long, device=x.device)
        return (c, c) if self.quantize_interface else c
