"""VQGAN — host-side mirror of the inference path of reference mebt/vqgan.py (`VQGAN.encode`, `VQGAN.decode`, :82-93) on the HIP
operators of csrc/vqgan.hip.  Same constructor argument (`args` namespace with n_hiddens / downsample / image_channels /
embedding_dim / n_codes), module tree and state-dict names (`encoder.conv_first.conv.weight`, `encoder.conv_blocks.0.res.norm1.weight`,
`decoder.conv_blocks.0.up.convt.weight`, `pre_vq_conv.conv.weight`, `codebook.embeddings`, ...) so reference checkpoints load;
the discriminators / LPIPS / losses (training of the first stage, SURVEY.md §2 OUT-OF-SCOPE) are not built and their keys are
ignored on load.

The nn.Modules below only HOLD parameters.  `encode` / `decode` walk a launch plan: every convolution is one
`mebt_op_conv3d` call per output sub-lattice (1 for a convolution, 4 or 8 for a stride-2 transposed convolution), every
Normalize + SiLU one `mebt_op_groupnorm_silu`, the codebook search one `mebt_op_codebook_argmin`.  Activations stay
channels-last on the GPU; `compute_dtype` = "f16" (MFMA, BASELINE config 5) or "f32" (parity).  Weights are re-laid out
([Cout][tap][Cin]) and cast once, at the first call after a (re)load.  No CPU / eager-torch compute path exists.
"""
import ctypes as C
import math

import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr, cur_stream


def _strides(downsample):
    n = np.array([int(math.log2(d)) for d in downsample])
    out = []
    for _ in range(int(n.max())):
        out.append(tuple(2 if d > 0 else 1 for d in n))                      # vqgan.py:277,320
        n = n - 1
    return out


class SamePadConv3d(nn.Module):
    """parameters of reference vqgan.py:374-398 (`conv` = nn.Conv3d, padding handled by the kernel's clamped gather)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True, padding_type='replicate'):
        super().__init__()
        k = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        s = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        if padding_type != 'replicate':
            raise NotImplementedError("only replicate padding (the reference default) is built")
        self.kernel_size, self.stride = k, s
        self.conv = nn.Conv3d(in_channels, out_channels, k, stride=s, padding=0, bias=bias)


class SamePadConvTranspose3d(nn.Module):
    """parameters of reference vqgan.py:401-424 (`convt` = nn.ConvTranspose3d)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True, padding_type='replicate'):
        super().__init__()
        k = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        s = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        self.kernel_size, self.stride = k, s
        self.convt = nn.ConvTranspose3d(in_channels, out_channels, k, stride=s, bias=bias, padding=tuple(kk - 1 for kk in k))


def Normalize(in_channels, norm_type='group'):
    if norm_type != 'group':
        raise NotImplementedError("only GroupNorm (norm_type='group', the reference default) is built")
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)    # vqgan.py:258


class SiLU(nn.Module):                         # placeholder so that `final_block.0` keeps its index (vqgan.py:285-288)
    pass


class ResBlock(nn.Module):
    """parameters of reference vqgan.py:338-355 (in_channels == out_channels everywhere in Encoder / Decoder)"""

    def __init__(self, in_channels, out_channels=None, norm_type='group', padding_type='replicate'):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        if in_channels != out_channels:
            raise NotImplementedError("ResBlock with a channel change (conv_shortcut) does not occur in the VQGAN")
        self.norm1 = Normalize(in_channels, norm_type)
        self.conv1 = SamePadConv3d(in_channels, out_channels, kernel_size=3, padding_type=padding_type)
        self.norm2 = Normalize(in_channels, norm_type)
        self.conv2 = SamePadConv3d(out_channels, out_channels, kernel_size=3, padding_type=padding_type)


class Encoder(nn.Module):
    def __init__(self, n_hiddens, downsample, image_channel=3, norm_type='group', padding_type='replicate'):
        super().__init__()
        self.conv_blocks = nn.ModuleList()
        self.conv_first = SamePadConv3d(image_channel, n_hiddens, kernel_size=3, padding_type=padding_type)
        out_channels = n_hiddens
        for i, stride in enumerate(_strides(downsample)):
            block = nn.Module()
            in_channels, out_channels = n_hiddens * 2 ** i, n_hiddens * 2 ** (i + 1)
            block.down = SamePadConv3d(in_channels, out_channels, 4, stride=stride, padding_type=padding_type)
            block.res = ResBlock(out_channels, out_channels, norm_type=norm_type)
            self.conv_blocks.append(block)
        self.final_block = nn.Sequential(Normalize(out_channels, norm_type), SiLU())
        self.out_channels = out_channels


class Decoder(nn.Module):
    def __init__(self, n_hiddens, upsample, image_channel, norm_type='group'):
        super().__init__()
        strides = _strides(upsample)
        max_us = len(strides)
        in_channels = n_hiddens * 2 ** max_us
        self.final_block = nn.Sequential(Normalize(in_channels, norm_type), SiLU())
        self.conv_blocks = nn.ModuleList()
        out_channels = in_channels
        for i, us in enumerate(strides):
            block = nn.Module()
            in_channels = in_channels if i == 0 else n_hiddens * 2 ** (max_us - i + 1)
            out_channels = n_hiddens * 2 ** (max_us - i)
            block.up = SamePadConvTranspose3d(in_channels, out_channels, 4, stride=us)
            block.res1 = ResBlock(out_channels, out_channels, norm_type=norm_type)
            block.res2 = ResBlock(out_channels, out_channels, norm_type=norm_type)
            self.conv_blocks.append(block)
        self.conv_last = SamePadConv3d(out_channels, image_channel, kernel_size=3)


class Codebook(nn.Module):
    """buffers of reference modules/codebook.py:13-24; inference only (no EMA update / random restart)"""

    def __init__(self, n_codes, embedding_dim, no_random_restart=False, restart_thres=1.0):
        super().__init__()
        self.register_buffer('embeddings', torch.randn(n_codes, embedding_dim))
        self.register_buffer('N', torch.zeros(n_codes))
        self.register_buffer('z_avg', self.embeddings.data.clone())
        self.n_codes, self.embedding_dim = n_codes, embedding_dim
        self._need_init = False

    def dictionary_lookup(self, encodings):
        """codebook.py:99-101 -> [..., embedding_dim] fp32"""
        ids = encodings.reshape(-1).contiguous()
        out = torch.empty(ids.numel(), self.embedding_dim, device=ids.device, dtype=torch.float32)
        check(_lib.load().mebt_op_embedding_rows(_lib.F32, ptr(ids), ptr(self.embeddings), ptr(out), ids.numel(), self.embedding_dim,
                                                 self.n_codes, cur_stream()))
        return out.view(*encodings.shape, self.embedding_dim)


class _Conv:
    """one prepared convolution: re-laid-out weights of every output sub-lattice + the integer geometry"""

    def __init__(self, weight, bias, kernel, stride, transposed, dtype):
        dev = weight.device
        tdt = torch.float16 if dtype == "f16" else torch.float32
        self.bias = bias.detach().to(dev, torch.float32).contiguous() if bias is not None else None
        self.transposed, self.kernel, self.stride = transposed, kernel, stride
        self.classes = []            # (pi, taps [n,3] int, weights [Cout, n, Cin])
        w = weight.detach().to(torch.float32)
        if not transposed:
            self.cout, self.cin = w.shape[0], w.shape[1]
            pad_front = [(k - s) // 2 + (k - s) % 2 for k, s in zip(kernel, stride)]                     # vqgan.py:385-388
            taps = [(kt - pad_front[0], kh - pad_front[1], kw - pad_front[2]) for kt in range(kernel[0]) for kh in range(kernel[1])
                    for kw in range(kernel[2])]
            wr = w.permute(0, 2, 3, 4, 1).reshape(self.cout, len(taps), self.cin)                       # [Cout][tap][Cin]
            self.classes.append(((0, 0, 0), taps, wr.to(tdt).contiguous()))
            self.os, self.sm = (1, 1, 1), stride
        else:
            self.cin, self.cout = w.shape[0], w.shape[1]
            # out o = i_p * s - (k - 1) + kk over the replicate-padded input i_p (pad front = (k-s)//2 + (k-s)%2); per dimension
            # and output parity pi (stride 2) only kk == (pi + k - 1) mod 2 contribute: i = o' + (pi + k - 1 - kk) / 2 - pad_front
            per_dim = []
            for k, s in zip(kernel, stride):
                pf = (k - s) // 2 + (k - s) % 2
                cls = {}
                for pi in range(s):
                    lst = []
                    for kk in range(k):
                        num = pi + (k - 1) - kk
                        if num % s == 0:
                            lst.append((kk, num // s - pf))
                    cls[pi] = lst
                per_dim.append(cls)
            for pt in range(stride[0]):
                for ph in range(stride[1]):
                    for pw in range(stride[2]):
                        taps, sel = [], []
                        for kt, ot in per_dim[0][pt]:
                            for kh, oh in per_dim[1][ph]:
                                for kw, ow in per_dim[2][pw]:
                                    taps.append((ot, oh, ow))
                                    sel.append((kt, kh, kw))
                        idx = torch.tensor(sel, device=dev)
                        wc = w[:, :, idx[:, 0], idx[:, 1], idx[:, 2]]                                    # [Cin, Cout, n]
                        self.classes.append(((pt, ph, pw), taps, wc.permute(1, 2, 0).to(tdt).contiguous()))
            self.os, self.sm = stride, (1, 1, 1)

    def out_dims(self, dims):
        if self.transposed:
            return tuple(d * s for d, s in zip(dims, self.stride))
        return tuple(d // s for d, s in zip(dims, self.stride))


class _Desc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("resid", C.c_void_p),
                ("B", C.c_int32), ("Ti", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Cin", C.c_int32),
                ("To", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
                ("cT", C.c_int32), ("cH", C.c_int32), ("cW", C.c_int32),
                ("os", C.c_int32 * 3), ("pi", C.c_int32 * 3), ("sm", C.c_int32 * 3), ("ntaps", C.c_int32),
                ("tap", (C.c_int8 * 4) * 64), ("in_mode", C.c_int32), ("out_mode", C.c_int32)]


class VQGAN(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.embedding_dim, self.n_codes = args.embedding_dim, args.n_codes
        norm_type = getattr(args, "norm_type", "group")
        padding_type = getattr(args, "padding_type", "replicate")
        image_channels = getattr(args, "image_channels", 3)
        self.encoder = Encoder(args.n_hiddens, args.downsample, image_channels, norm_type, padding_type)
        self.decoder = Decoder(args.n_hiddens, args.downsample, image_channels, norm_type)
        self.enc_out_ch = self.encoder.out_channels
        self.pre_vq_conv = SamePadConv3d(self.enc_out_ch, args.embedding_dim, 1, padding_type=padding_type)
        self.post_vq_conv = SamePadConv3d(args.embedding_dim, self.enc_out_ch, 1)
        self.codebook = Codebook(args.n_codes, args.embedding_dim)
        self.compute_dtype = "f16"
        self.use_mfma = True
        self._prepared = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: setattr(module, "_prepared", None))

    @property
    def latent_shape(self):
        a = self.args
        shape = (a.sequence_length // a.sample_every_n_frames, a.resolution, a.resolution)
        return tuple(s // d for s, d in zip(shape, a.downsample))

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu"):
        """Lightning checkpoint of the reference (download.py:56-61): {'state_dict', 'hyper_parameters': {'args': Namespace}}"""
        ck = torch.load(path, map_location=map_location, weights_only=False)
        hp = ck.get("hyper_parameters", {})
        model = cls(hp["args"] if "args" in hp else hp)
        model.load_state_dict(ck["state_dict"], strict=False)
        return model

    def load_state_dict(self, state_dict, strict=True, **kw):
        """discriminator / perceptual-loss tensors of a reference checkpoint are not part of the inference path"""
        skip = ("image_discriminator.", "video_discriminator.", "perceptual_model.")
        sd = {k: v for k, v in state_dict.items() if not k.startswith(skip)}
        return super().load_state_dict(sd, strict=strict, **kw)

    # ---- preparation ---------------------------------------------------------------------------------------------------
    def _prepare(self):
        dev = self.codebook.embeddings.device
        if dev.type != "cuda":
            raise RuntimeError("mebt_amd.vqgan runs on MI355X only: move the model to the GPU (there is no CPU path in the product)")
        key = (self.compute_dtype, dev)
        if self._prepared is not None and self._prepared["key"] == key:
            return self._prepared
        dt = self.compute_dtype
        prep = {"key": key, "convs": {}}

        def conv(mod):
            if isinstance(mod, SamePadConv3d):
                return _Conv(mod.conv.weight, mod.conv.bias, mod.kernel_size, mod.stride, False, dt)
            return _Conv(mod.convt.weight, mod.convt.bias, mod.kernel_size, mod.stride, True, dt)

        for name, mod in self.named_modules():
            if isinstance(mod, (SamePadConv3d, SamePadConvTranspose3d)):
                prep["convs"][name] = conv(mod)
        prep["emb"] = self.codebook.embeddings.detach().to(torch.float32).contiguous()
        self._prepared = prep
        return prep

    # ---- launch helpers ----------------------------------------------------------------------------------------------------
    def _tdt(self):
        return torch.float16 if self.compute_dtype == "f16" else torch.float32

    def _code(self):
        return _lib.F16 if self.compute_dtype == "f16" else _lib.F32

    def _conv(self, name, x, B, dims, resid=None, in_mode=0, out_mode=0):
        """x: channels-last [B, *dims, Cin] (or the fp32 video when in_mode = 1) -> (out, out_dims)"""
        cv = self._prepared["convs"][name]
        od = cv.out_dims(dims)
        dev = x.device
        if out_mode == 0:
            out = torch.empty(B, *od, cv.cout, device=dev, dtype=self._tdt())
        elif out_mode == 1:
            out = torch.empty(B, *od, cv.cout, device=dev, dtype=torch.float32)
        else:
            out = torch.empty(B, cv.cout, *od, device=dev, dtype=torch.float32)
        lib = _lib.load()
        for pi, taps, w in cv.classes:
            d = _Desc()
            d.in_, d.out, d.w, d.bias, d.resid = ptr(x), ptr(out), ptr(w), ptr(cv.bias), ptr(resid)
            d.B, d.Ti, d.Hi, d.Wi, d.Cin = B, dims[0], dims[1], dims[2], cv.cin
            d.To, d.Ho, d.Wo, d.Cout = od[0], od[1], od[2], cv.cout
            cls = [(o - p + s - 1) // s for o, p, s in zip(od, pi, cv.os)]
            d.cT, d.cH, d.cW = cls
            for k in range(3):
                d.os[k], d.pi[k], d.sm[k] = cv.os[k], pi[k], cv.sm[k]
            d.ntaps = len(taps)
            for j, t in enumerate(taps):
                d.tap[j][0], d.tap[j][1], d.tap[j][2] = t
            d.in_mode, d.out_mode = in_mode, out_mode
            check(lib.mebt_op_conv3d(self._code(), C.byref(d), 1 if self.use_mfma else 0, cur_stream()))
        return out, od

    def _norm_silu(self, gn, x, B, dims):
        Cc = x.shape[-1]
        y = torch.empty_like(x)
        stats = torch.empty(B * 64, device=x.device, dtype=torch.float32)
        g, b = gn.weight.detach().float().contiguous(), gn.bias.detach().float().contiguous()
        check(_lib.load().mebt_op_groupnorm_silu(self._code(), ptr(x), ptr(y), ptr(g), ptr(b), ptr(stats), B, int(np.prod(dims)), Cc,
                                                 cur_stream()))
        return y

    def _res(self, prefix, mod, x, B, dims):
        """ResBlock.forward (vqgan.py:357-370): x + conv2(silu(norm2(conv1(silu(norm1(x))))))"""
        h = self._norm_silu(mod.norm1, x, B, dims)
        h, _ = self._conv(prefix + ".conv1", h, B, dims)
        h = self._norm_silu(mod.norm2, h, B, dims)
        h, _ = self._conv(prefix + ".conv2", h, B, dims, resid=x)
        return h

    # ---- reference API ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x, include_embeddings=False):
        """VQGAN.encode (vqgan.py:82-88): x [B, C, T, H, W] float -> encodings [B, t, h, w] int64 (and, with include_embeddings,
        the quantised embeddings [B, c, t, h, w] first)"""
        self._prepare()
        x = x.to(torch.float32).contiguous()
        B, dims = x.shape[0], tuple(x.shape[2:])
        h, dims = self._conv("encoder.conv_first", x, B, dims, in_mode=1)
        for i, blk in enumerate(self.encoder.conv_blocks):
            h, dims = self._conv(f"encoder.conv_blocks.{i}.down", h, B, dims)
            h = self._res(f"encoder.conv_blocks.{i}.res", blk.res, h, B, dims)
        h = self._norm_silu(self.encoder.final_block[0], h, B, dims)
        z, dims = self._conv("pre_vq_conv", h, B, dims, out_mode=1)                    # fp32 [B, t, h, w, c] for the search
        M = B * int(np.prod(dims))
        emb = self._prepared["emb"]
        esq = torch.empty(self.n_codes + 1, device=x.device, dtype=torch.float32)
        ids = torch.empty(M, device=x.device, dtype=torch.long)
        # large searches (config 5 at batch 16: 16384 x 16384 x 256): approximate scores on the bf16 MFMA GEMM + exact fp32 re-evaluation
        # of every code that can still be the arg-min (csrc/vqgan.hip codebook_filter_kernel); small ones: the exact fp32 GEMM directly
        if (self.use_mfma and self.embedding_dim % 64 == 0 and self.n_codes % 8 == 0 and M * self.n_codes >= (1 << 24)
                and os.environ.get("MEBT_CODEBOOK_FILTER", "1") != "0"):
            lowp = torch.empty((M + self.n_codes) * self.embedding_dim, device=x.device, dtype=torch.bfloat16)
            score = torch.empty(M, self.n_codes, device=x.device, dtype=torch.bfloat16)     # approximate scores: half the bytes of the exact path's matrix
            check(_lib.load().mebt_op_codebook_argmin_filtered(ptr(z), ptr(emb), ptr(score), ptr(esq), ptr(lowp), ptr(ids), M, self.n_codes,
                                                               self.embedding_dim, cur_stream()))
        else:
            score = torch.empty(M, self.n_codes, device=x.device, dtype=torch.float32)
            check(_lib.load().mebt_op_codebook_argmin(ptr(z), ptr(emb), ptr(score), ptr(esq), ptr(ids), M, self.n_codes, self.embedding_dim,
                                                      cur_stream()))
        self._last_z = z
        ids = ids.view(B, *dims)
        if include_embeddings:
            e = self.codebook.dictionary_lookup(ids)                                     # [B, t, h, w, c]
            return e.permute(0, 4, 1, 2, 3), ids                                         # a view: [B, c, t, h, w] (shift_dim, codebook.py:63)
        return ids

    @torch.no_grad()
    def decode(self, encodings):
        """VQGAN.decode (vqgan.py:90-93): encodings [B, t, h, w] int64 -> video [B, C, T, H, W] fp32"""
        self._prepare()
        ids = encodings.contiguous()
        B, dims = ids.shape[0], tuple(ids.shape[1:])
        rows = ids.numel()
        h = torch.empty(B, *dims, self.embedding_dim, device=ids.device, dtype=self._tdt())
        check(_lib.load().mebt_op_embedding_rows(self._code(), ptr(ids), ptr(self._prepared["emb"]), ptr(h), rows, self.embedding_dim,
                                                 self.n_codes, cur_stream()))
        h, dims = self._conv("post_vq_conv", h, B, dims)
        h = self._norm_silu(self.decoder.final_block[0], h, B, dims)
        for i, blk in enumerate(self.decoder.conv_blocks):
            h, dims = self._conv(f"decoder.conv_blocks.{i}.up", h, B, dims)
            h = self._res(f"decoder.conv_blocks.{i}.res1", blk.res1, h, B, dims)
            h = self._res(f"decoder.conv_blocks.{i}.res2", blk.res2, h, B, dims)
        out, _ = self._conv("decoder.conv_last", h, B, dims, out_mode=2)
        return out

    def forward(self, x):
        """reconstruction (the inference part of vqgan.py:95-100): decode(encode(x))"""
        return self.decode(self.encode(x))


def load_vqgan(vqgan_ckpt, device=torch.device('cpu')):
    """reference download.py:50-54"""
    vqgan = VQGAN.load_from_checkpoint(vqgan_ckpt).to(device)
    vqgan.eval()
    return vqgan
