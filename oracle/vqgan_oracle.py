"""CPU oracle for the 3D-VQGAN first stage (BASELINE.json configs[4]): a from-scratch functional
restatement of the reference's inference path `VQGAN.encode` / `VQGAN.decode` in plain fp32 PyTorch on the CPU.

TEST INFRASTRUCTURE ONLY (same rules as oracle/mebt_oracle.py): imported by tests/, never by the product.

Parity pin: tests/golden/make_golden.py imports the real reference `mebt.vqgan.VQGAN` (LPIPS and the third-party modules
stubbed, closed-form weights) and commits inputs + reduced outputs in tests/golden/vqgan_*.npz; tests/test_oracle_golden.py
checks the functions below against them.

`P` is a dict {reference state-dict name -> fp32 tensor}: `encoder.conv_first.conv.weight`, `encoder.conv_blocks.0.down.conv.weight`,
`encoder.conv_blocks.0.res.norm1.weight`, ..., `decoder.conv_blocks.0.up.convt.weight`, `pre_vq_conv.conv.weight`,
`post_vq_conv.conv.weight`, `codebook.embeddings` (paths relative to /root/reference are cited per function).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

GN_EPS = 1e-6          # mebt/vqgan.py:258 (GroupNorm(32, C, eps=1e-6))
GN_GROUPS = 32


class VQGANConfig:
    """mebt/vqgan.py:228-249 (argparse names); TATS-style values for BASELINE config 5 are the defaults here."""

    def __init__(self, n_hiddens=32, downsample=(4, 8, 8), image_channels=3, embedding_dim=256, n_codes=16384):
        self.n_hiddens, self.downsample = n_hiddens, tuple(downsample)
        self.image_channels, self.embedding_dim, self.n_codes = image_channels, embedding_dim, n_codes

    @property
    def n_times(self):
        return [int(math.log2(d)) for d in self.downsample]


def _strides(n_times):
    """per-level stride tuples of Encoder / Decoder (vqgan.py:271-283, 313-324): 2 while a dimension still has
    down/up-sampling to do, else 1"""
    n = np.array(n_times)
    out = []
    for _ in range(int(n.max())):
        out.append(tuple(2 if d > 0 else 1 for d in n))
        n = n - 1
    return out


def param_shapes(cfg):
    """state-dict schema of the inference path (encoder, decoder, pre/post convs, codebook)"""
    h, C = cfg.n_hiddens, cfg.image_channels
    s = {"encoder.conv_first.conv.weight": (h, C, 3, 3, 3), "encoder.conv_first.conv.bias": (h,)}

    def res(prefix, c):
        for k in ("norm1", "norm2"):
            s[f"{prefix}.{k}.weight"] = (c,)
            s[f"{prefix}.{k}.bias"] = (c,)
        for k in ("conv1", "conv2"):
            s[f"{prefix}.{k}.conv.weight"] = (c, c, 3, 3, 3)
            s[f"{prefix}.{k}.conv.bias"] = (c,)

    strides = _strides(cfg.n_times)
    max_s = len(strides)
    out_c = h
    for i in range(max_s):
        in_c, out_c = h * 2 ** i, h * 2 ** (i + 1)
        s[f"encoder.conv_blocks.{i}.down.conv.weight"] = (out_c, in_c, 4, 4, 4)
        s[f"encoder.conv_blocks.{i}.down.conv.bias"] = (out_c,)
        res(f"encoder.conv_blocks.{i}.res", out_c)
    s["encoder.final_block.0.weight"] = (out_c,)
    s["encoder.final_block.0.bias"] = (out_c,)
    enc_out = out_c
    s["pre_vq_conv.conv.weight"] = (cfg.embedding_dim, enc_out, 1, 1, 1)
    s["pre_vq_conv.conv.bias"] = (cfg.embedding_dim,)
    s["post_vq_conv.conv.weight"] = (enc_out, cfg.embedding_dim, 1, 1, 1)
    s["post_vq_conv.conv.bias"] = (enc_out,)
    s["codebook.embeddings"] = (cfg.n_codes, cfg.embedding_dim)
    in_c = h * 2 ** max_s
    s["decoder.final_block.0.weight"] = (in_c,)
    s["decoder.final_block.0.bias"] = (in_c,)
    for i in range(max_s):
        ic = in_c if i == 0 else h * 2 ** (max_s - i + 1)
        oc = h * 2 ** (max_s - i)
        s[f"decoder.conv_blocks.{i}.up.convt.weight"] = (ic, oc, 4, 4, 4)          # ConvTranspose3d: [in, out, k, k, k]
        s[f"decoder.conv_blocks.{i}.up.convt.bias"] = (oc,)
        res(f"decoder.conv_blocks.{i}.res1", oc)
        res(f"decoder.conv_blocks.{i}.res2", oc)
    s["decoder.conv_last.conv.weight"] = (C, oc, 3, 3, 3)
    s["decoder.conv_last.conv.bias"] = (C,)
    return s


def closed_form_params(cfg):
    """deterministic machine-independent weights: conv weights ~ N(0, 1/fan_in) (activations stay O(1) through the stack),
    small biases, GroupNorm gains around 1, codebook ~ N(0, 1) like its `torch.randn` initialisation (codebook.py:15)"""
    from oracle import closed_form as cf
    P = {}
    for name, shape in param_shapes(cfg).items():
        if name == "codebook.embeddings":
            a = cf.pseudo_normal("vqgan/" + name, shape, std=1.0)
        elif name.endswith("convt.weight"):
            a = cf.pseudo_normal("vqgan/" + name, shape, std=1.0 / math.sqrt(shape[0] * 8))     # 8 of the 64 taps reach an output voxel at stride 2
        elif name.endswith("conv.weight"):
            a = cf.pseudo_normal("vqgan/" + name, shape, std=1.0 / math.sqrt(int(np.prod(shape[1:]))))
        elif ".norm" in name or "final_block" in name:
            a = cf.pseudo_normal("vqgan/" + name, shape, std=0.1) + (1.0 if name.endswith("weight") else 0.0)
        else:
            a = cf.pseudo_normal("vqgan/" + name, shape, std=0.05)
        P[name] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return P


# ---- layers ----------------------------------------------------------------------------------------------------------------
def _same_pad(kernel, stride):
    """SamePadConv3d / SamePadConvTranspose3d (vqgan.py:378-391, 402-412): total pad k - s per dimension, the larger half in
    front; F.pad order is (W_front, W_back, H_front, H_back, T_front, T_back)"""
    pad = []
    for k, s in zip(kernel[::-1], stride[::-1]):
        p = k - s
        pad += [p // 2 + p % 2, p // 2]
    return tuple(pad)


def same_pad_conv3d(x, w, b, stride=(1, 1, 1)):
    """vqgan.py:374-398: replicate-pad then Conv3d(padding=0)"""
    k = tuple(w.shape[2:])
    return F.conv3d(F.pad(x, _same_pad(k, stride), mode="replicate"), w, b, stride=stride)


def same_pad_conv_transpose3d(x, w, b, stride):
    """vqgan.py:401-424: replicate-pad then ConvTranspose3d(padding = k - 1)"""
    k = tuple(w.shape[2:])
    return F.conv_transpose3d(F.pad(x, _same_pad(k, stride), mode="replicate"), w, b, stride=stride, padding=tuple(kk - 1 for kk in k))


def norm_silu(x, w, b):
    """Normalize (GroupNorm 32 groups, eps 1e-6, vqgan.py:255-258) followed by x * sigmoid(x) (:17-18)"""
    h = F.group_norm(x, GN_GROUPS, w, b, eps=GN_EPS)
    return h * torch.sigmoid(h)


def res_block(P, pre, x):
    """ResBlock.forward (vqgan.py:357-370) with in_channels == out_channels (every block of Encoder / Decoder); note that
    norm2 is built on in_channels (:351) — the same thing here"""
    h = norm_silu(x, P[pre + ".norm1.weight"], P[pre + ".norm1.bias"])
    h = same_pad_conv3d(h, P[pre + ".conv1.conv.weight"], P[pre + ".conv1.conv.bias"])
    h = norm_silu(h, P[pre + ".norm2.weight"], P[pre + ".norm2.bias"])
    h = same_pad_conv3d(h, P[pre + ".conv2.conv.weight"], P[pre + ".conv2.conv.bias"])
    return x + h


def encoder(P, cfg, x):
    """Encoder.forward (vqgan.py:290-296)"""
    h = same_pad_conv3d(x, P["encoder.conv_first.conv.weight"], P["encoder.conv_first.conv.bias"])
    for i, st in enumerate(_strides(cfg.n_times)):
        h = same_pad_conv3d(h, P[f"encoder.conv_blocks.{i}.down.conv.weight"], P[f"encoder.conv_blocks.{i}.down.conv.bias"], stride=st)
        h = res_block(P, f"encoder.conv_blocks.{i}.res", h)
    return norm_silu(h, P["encoder.final_block.0.weight"], P["encoder.final_block.0.bias"])


def codebook_distances(z, emb):
    """Codebook.forward (modules/codebook.py:52-56): |z|^2 - 2 z E^T + |E|^2 on the flattened [b t h w, c] inputs"""
    flat = z.permute(0, 2, 3, 4, 1).reshape(-1, z.shape[1])
    return (flat ** 2).sum(dim=1, keepdim=True) - 2 * flat @ emb.t() + (emb.t() ** 2).sum(dim=0, keepdim=True)


def encode(P, cfg, x, return_all=False):
    """VQGAN.encode (vqgan.py:82-88): encoder -> pre_vq_conv -> nearest codebook entry.  -> encodings [b, t, h, w] int64
    (and with return_all: the pre-quantisation z [b, c, t, h, w] and the distance matrix)"""
    z = same_pad_conv3d(encoder(P, cfg, x), P["pre_vq_conv.conv.weight"], P["pre_vq_conv.conv.bias"])
    d = codebook_distances(z, P["codebook.embeddings"])
    ids = torch.argmin(d, dim=1).view(z.shape[0], *z.shape[2:])               # codebook.py:58,60
    return (ids, z, d) if return_all else ids


def decoder(P, cfg, h):
    """Decoder.forward (vqgan.py:328-335)"""
    h = norm_silu(h, P["decoder.final_block.0.weight"], P["decoder.final_block.0.bias"])
    for i, st in enumerate(_strides(cfg.n_times)):
        h = same_pad_conv_transpose3d(h, P[f"decoder.conv_blocks.{i}.up.convt.weight"], P[f"decoder.conv_blocks.{i}.up.convt.bias"], st)
        h = res_block(P, f"decoder.conv_blocks.{i}.res1", h)
        h = res_block(P, f"decoder.conv_blocks.{i}.res2", h)
    return same_pad_conv3d(h, P["decoder.conv_last.conv.weight"], P["decoder.conv_last.conv.bias"])


def decode(P, cfg, ids):
    """VQGAN.decode (vqgan.py:90-93): embedding lookup -> [b, c, t, h, w] -> post_vq_conv -> decoder"""
    h = F.embedding(ids, P["codebook.embeddings"]).permute(0, 4, 1, 2, 3)
    h = same_pad_conv3d(h, P["post_vq_conv.conv.weight"], P["post_vq_conv.conv.bias"])
    return decoder(P, cfg, h)
