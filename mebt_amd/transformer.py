"""Net2NetTransformer — host-side mirror of reference mebt/transformer.py:60-824 on top of the HIP
engine.  Same constructor, attributes (`transformer`, `mask_sampler`, `tok_emb`, `mask_emb`,
`sos_emb`, `pos_emb`, `first_stage_model`, `config`, `learning_rate` ...), state-dict names and
method signatures, so `train_transformer.py` / `draft_and_revise_videos.py` style callers work
unchanged.  Written from scratch: every tensor op of the reference is replaced by calls into
libmebt_hip.so (mebt_amd/_lib.py); PyTorch only owns memory and streams.

Precision: `compute_dtype` = "bf16" (MFMA bf16, fp32 accumulate — what bench.py measures) or
"f32" (exact-fp32 MFMA — the 1e-3 parity mode).  Set it before the first forward, or through the
MEBT_COMPUTE_DTYPE environment variable.
"""
import copy
import math
import os
import random
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .config import instantiate_from_config
from .engine import NativeModel, flat_layout
from .lightning_shim import LightningModuleShim
from .modules.gpt import GPT
from .modules.encoders import SOSProvider


# ---- video-length priors (reference transformer.py:25-49; resolved there with eval(name)) ----------
def uniform(vid_lengths, t):
    return np.ones_like(vid_lengths, dtype=float)


def _gauss(vid_lengths, t, b, c):
    return np.exp((-(t - (vid_lengths - 1) * b) ** 2) / (2 * (b * c) ** 2))


def gaussian(vid_lengths, t, b, c):
    return _gauss(vid_lengths, t, b, c)


def gaussian100000_2(vid_lengths, t):
    return _gauss(vid_lengths, t, 100000, 2)


def gaussian2(vid_lengths, t):
    return _gauss(vid_lengths, t, 30000, 2)


def longest(vid_lengths, t):
    x = np.zeros_like(vid_lengths, dtype=float)
    x[-1] = 1.
    return x


T_PRIORS = {"uniform": uniform, "gaussian100000_2": gaussian100000_2, "gaussian2": gaussian2, "longest": longest}


# ---- context-temperature schedules (reference transformer.py:51-58) -----------------------------------
def linear(t):
    return 1. - t


def constant(t):
    return 1.


def cosine(t):
    return np.cos(t * np.pi / 2.)


CTEMP_SCHEDULES = {"linear": linear, "constant": constant, "cosine": cosine}


def disabled_train(self, mode=True):
    return self


def _cfg_has(cfg, key):
    return hasattr(cfg, key) if not isinstance(cfg, dict) else key in cfg


# ---- autograd bridges ----------------------------------------------------------------------------------
def _check_generation(model, generation):
    """The engine keeps ONE set of saved activations (the last training-mode forward): a backward that belongs to an
    earlier forward would silently use the wrong activations and indices."""
    if model._native.generation != generation:
        raise RuntimeError("mebt_amd: backward of a forward whose activations were overwritten by a later training-mode "
                           "forward (the engine keeps one set of saved activations: run forward -> backward pairs in order; "
                           "for gradient accumulation use TrainLoop(accumulate_grad_batches=k))")


class _LogitsFn(torch.autograd.Function):
    """logits = engine.forward(...); backward feeds dL/dlogits to the HIP backward, which writes the
    parameter gradients straight into the flat gradient buffers (`param.grad` are views of them)."""

    @staticmethod
    def forward(ctx, trigger, model, x_ids, ci, ti):
        ctx.model = model
        out = model._native.forward(x_ids, ci, ti, training=True, dropout_seed=model._next_seed())
        ctx.generation = model._native.generation
        return out

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.model
        _check_generation(m, ctx.generation)
        m._native.backward(None, 0.0, between=m._bucket_hook, dlogits=dlogits)
        m._attach_grads()
        return torch.zeros((), device=dlogits.device), None, None, None, None


class _GptFn(torch.autograd.Function):
    """GPT.forward on caller-embedded tensors with autograd: gradients flow to the three inputs and into the flat
    parameter-gradient buffers (reference gpt.py:234-253 under torch autograd)."""

    @staticmethod
    def forward(ctx, model, sos, contexts, targets):
        nm = model._native
        ctx.model = model
        ctx.shapes = (tuple(sos.shape), tuple(contexts.shape), tuple(targets.shape))
        out = nm.gpt_forward_train(sos, contexts, targets, dropout_seed=model._next_seed(), dropout=model.transformer.training)
        ctx.generation = nm.generation
        return out

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.model
        _check_generation(m, ctx.generation)
        ds, dc, dt = m._native.gpt_backward(dlogits, ctx.shapes)
        m._attach_grads()
        return None, ds, dc, dt


class _LossFn(torch.autograd.Function):
    """Fused masked-token loss (+top-1/top-5) on the logits of the last training forward; the
    backward runs the fused CE-backward kernel and the whole network backward."""

    @staticmethod
    def forward(ctx, trigger, model, logits, scale):
        ctx.model, ctx.logits, ctx.scale = model, logits, scale
        ctx.generation = model._native.generation
        stats = model._native.loss_stats(logits)
        ctx.mark_non_differentiable(stats)
        return (stats[0] * scale).to(torch.float32), stats

    @staticmethod
    def backward(ctx, gloss, _gstats):
        m = ctx.model
        _check_generation(m, ctx.generation)
        up = gloss.to(torch.float32).contiguous()
        m._native.backward(ctx.logits, ctx.scale, upstream=up, between=m._bucket_hook)
        m._attach_grads()
        return torch.zeros((), device=gloss.device), None, None, None


class MebtAdamW(torch.optim.Optimizer):
    """torch.optim.Optimizer façade over the fused AdamW kernel: keeps the 4 param_groups of
    reference transformer.py:790-797 (so `optimizer_step`'s per-group LR writes work) but updates
    the two flat buffers with one kernel family (mebt_adamw_step)."""

    def __init__(self, model, groups, lr, betas=(0.9, 0.95), eps=1e-8):
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0))
        self._model = weakref.ref(model)
        self._steps = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        m = self._model()
        m._ensure_native()
        lrs = {g["lr"] for g in self.param_groups}
        if len(lrs) != 1:
            raise NotImplementedError("per-group learning rates differ; the reference sets one LR for all groups")
        self._steps += 1
        g0 = self.param_groups[0]
        red = m._reducer
        m._native.adamw_step(g0["lr"], g0["weight_decay"], self._steps, betas=g0["betas"], eps=g0["eps"],
                             grad_scale=(red.grad_scale if red is not None else 1.0))
        return loss

    def zero_grad(self, set_to_none=False):
        pass        # every backward overwrites the gradient buffers

    # the moments live in the engine's flat buffers (not in self.state): persist them and the step count
    def state_dict(self):
        m = self._model()
        nm = m._ensure_native()
        return {"mebt_flat_adam": [t.detach().cpu().clone() for t in nm._adam_state()], "steps": self._steps,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        m = self._model()
        nm = m._ensure_native()
        for dst, src in zip(nm._adam_state(), sd["mebt_flat_adam"]):
            dst.copy_(src)
        self._steps = int(sd["steps"])
        for g, saved in zip(self.param_groups, sd.get("param_groups", [])):
            g.update(saved)


class _KvSession:
    """One sampling loop's cache of the latent_enc blocks' key / value projections over all N positions (engine:
    mebt_forward_kvcache).  `contexts` is read-only through the network (reference modules/gpt.py:187-192), so a context position's
    K / V rows change only when its token id does — the loops (reference transformer.py:353-447, :544-663) re-sample a few hundred
    positions per forward.  Each forward names the positions to re-project (`dirty`); default = the previous forward's targets (the
    only ids that changed since), the first forward of a session re-projects every context position.
    MEBT_KV_CACHE_CHECK=1 (tests): a per-position record of the token id each cache row was computed for, checked against the ids
    at every context position of every forward (one host synchronisation per forward)."""

    def __init__(self, native, B, N):
        self.native, self.B, self.N = native, B, N
        self.cache = native.new_kv_cache(B, N)
        self.fresh = True
        self.degraded = False
        self.prev_targets = None
        self.check = os.environ.get("MEBT_KV_CACHE_CHECK", "0") == "1"
        self.stamp = torch.full((B, N), -1, dtype=torch.long, device=native.device) if self.check else None
        self.rows_projected = 0           # context rows re-projected so far (statistics for bench / tests)
        self.rows_uncached = 0            # ... and what the same forwards project without the cache

    def forward(self, x_ids, ci, ti, dirty, logits_bf16):
        # The engine takes 0 <= ND <= NC re-projected positions (its context buffers hold NC rows).  What must hold is that every
        # position of `ci` whose token changed since its row was written is in `d`:
        #   * nothing projected yet (`fresh`: a new session, or after a forward without contexts - a one-step draft / revise pass
        #     re-samples every position) -> the whole context;
        #   * the loop named the changed positions (`dirty`, a subset of ci) -> those;
        #   * otherwise the previous forward's targets, if that many rows fit;  if they do not, the whole context is projected and
        #     the session stays `degraded` (some changed positions outside `ci` were NOT re-projected, so no later `dirty` list can
        #     be trusted to cover them): every further forward re-projects its whole context, like the reference.
        NC = ci.shape[1]
        if self.fresh or self.degraded or NC == 0:
            d = ci
        elif dirty is not None:
            d = dirty
        elif self.prev_targets is not None and self.prev_targets.shape[1] <= NC:
            d = self.prev_targets
        else:
            d = ci
            self.degraded = True
        d = d.reshape(self.B, -1)
        if self.check:
            self.stamp.scatter_(1, d, x_ids.gather(1, d))
            if ci.shape[1]:
                stale = int((self.stamp.gather(1, ci) != x_ids.gather(1, ci)).sum())
                assert stale == 0, f"key/value cache: {stale} context positions were projected for another token id"
        logits = self.native.forward_cached(x_ids, ci, ti, self.cache, d, logits_bf16)
        self.fresh = NC == 0           # a forward without contexts projected nothing and its targets are every position it names
        self.prev_targets = ti
        self.rows_projected += self.B * d.shape[1]
        self.rows_uncached += self.B * ci.shape[1]
        return logits


class Net2NetTransformer(LightningModuleShim):
    def __init__(self, transformer_config, first_stage_config, mask_config, ckpt_path=None, ignore_keys=[],
                 first_stage_key="video", cond_stage_key="label", pkeep=1.0, sos_token=0):
        super().__init__()
        cfg = self.config = transformer_config
        self.class_cond_dim = cfg.class_cond_dim if _cfg_has(cfg, "class_cond_dim") else None
        self.be_unconditional = cfg.unconditional
        self.sos_token = sos_token
        self.first_stage_key = first_stage_key
        self.first_stage_vocab_size = cfg.vocab_size
        self.cond_stage_key = cond_stage_key
        self.vtokens = cfg.vtokens
        self.n_embd = cfg.n_embd
        self.vis_epoch = cfg.vis_epoch if _cfg_has(cfg, "vis_epoch") else 100
        # optional keys get in-place defaults, like reference :83-100
        cfg.avg_loss = float(cfg.avg_loss) if _cfg_has(cfg, "avg_loss") else 0.0
        for k in ("embd_pdrop", "resid_pdrop", "attn_pdrop"):
            if not _cfg_has(cfg, k):
                setattr(cfg, k, 0.0)
        self.sample_every_n_latent_frames = cfg.sample_every_n_latent_frames if _cfg_has(cfg, "sample_every_n_latent_frames") else 0
        self.label_smoothing = cfg.label_smoothing if _cfg_has(cfg, "label_smoothing") else 0.0

        self.init_first_stage_from_ckpt(first_stage_config)
        self.init_cond_stage_from_ckpt(cfg)
        gpt_vocab = self.first_stage_vocab_size + self.cond_stage_vocab_size
        self.transformer = GPT(gpt_vocab, cfg.block_size, n_layer=cfg.n_layer, n_head=cfg.n_head, n_embd=cfg.n_embd,
                               vtokens_pos=cfg.vtokens_pos if _cfg_has(cfg, "vtokens_pos") else False,
                               n_unmasked=cfg.n_unmasked if _cfg_has(cfg, "n_unmasked") else 0,
                               attn_pdrop=cfg.attn_pdrop, embd_pdrop=cfg.embd_pdrop, resid_pdrop=cfg.resid_pdrop,
                               mode=cfg.mode)
        self.transformer._owner = weakref.ref(self)
        self.mask_sampler = instantiate_from_config(config=mask_config)

        if not _cfg_has(cfg, "beta_params"):                                      # reference :113-119
            mp = mask_config.params if _cfg_has(mask_config, "params") else {}
            self.range = mp.t_range if _cfg_has(mp, "t_range") else [0., 1.]
            self.beta = False
        else:
            self.beta_params = cfg.beta_params
            self.beta_iter = float(cfg.beta_iter)
            self.beta = True
        self.t_lengths = np.array(list(range(self.mask_sampler.shape[0]))) + 1
        if not _cfg_has(cfg, "t_prior"):
            cfg.t_prior = "longest"
        if cfg.t_prior not in T_PRIORS:
            raise ValueError(f"unknown t_prior {cfg.t_prior!r} (known: {sorted(T_PRIORS)})")
        self.t_prior = T_PRIORS[cfg.t_prior]                                       # reference :125 uses eval()

        self.tok_emb = nn.Embedding(gpt_vocab, cfg.n_embd)
        self.tok_emb.weight.data.normal_(mean=0.0, std=0.02)
        self.mask_emb = nn.Parameter(torch.zeros(1, 1, cfg.n_embd))
        self.mask_emb.data.normal_(mean=0.0, std=0.02)
        if not _cfg_has(cfg, "sos_emb"):
            cfg.sos_emb = 1
        if cfg.sos_emb > 0:
            self.sos_emb = nn.Parameter(torch.zeros(1, cfg.sos_emb, cfg.n_embd))
            self.sos_emb.data.normal_(mean=0.0, std=0.02)
        self.num_pos = np.prod(self.mask_sampler.shape[1:])
        self.pos_emb = nn.Parameter(torch.zeros(1, cfg.block_size, cfg.n_embd))
        self.pos_emb.data.normal_(mean=0.0, std=0.02)
        self.n_head = cfg.n_head
        self.first_mode = cfg.mode[0]
        # training hyper-parameters the launcher sets post-hoc (train_transformer.py:54-66)
        self.learning_rate, self.warmup_steps, self.weight_decay, self.cosine_lr = 4.5e-6, 0, 0.01, False
        # engine state
        self.compute_dtype = os.environ.get("MEBT_COMPUTE_DTYPE", "bf16")
        self._native = None
        self._reducer = None            # mebt_amd.parallel.GradReducer when data-parallel
        self._seed_ctr = 0
        self.noise_hook = None          # tests inject Exp(1) noise: fn(kind, shape) -> tensor
        # weights loaded after the engine exists go straight into its flat fp32 buffer: refresh the bf16 mirror
        self.register_load_state_dict_post_hook(
            lambda module, incompatible: module._native.sync_lowp(force=True) if module._native is not None else None)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)
        self.pkeep = pkeep
        self.save_hyperparameters(transformer_config=transformer_config, first_stage_config=first_stage_config,
                                  mask_config=mask_config, first_stage_key=first_stage_key,
                                  cond_stage_key=cond_stage_key, pkeep=pkeep, sos_token=sos_token)

    # ---- construction helpers -----------------------------------------------------------------------
    def init_from_ckpt(self, path, ignore_keys=list()):
        """reference :170-178: load `state_dict`, dropping keys by prefix, strict=False"""
        sd = torch.load(path, map_location="cpu", weights_only=False)["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                print("Deleting key {} from state_dict.".format(k))
                del sd[k]
        self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}")

    def init_first_stage_from_ckpt(self, config):
        """reference :180-192.  `vtokens: True`: the batch carries token grids, no first stage, vocabulary forced to 16384.
        `vtokens: False`: the 3D-VQGAN first stage (mebt_amd/vqgan.py on the HIP operators) is loaded from
        `config.params.ckpt_path`, frozen, and its codebook size is the vocabulary; without a checkpoint path the model is
        still constructible (token grids only) and a first stage can be attached later (`model.first_stage_model = VQGAN(...)`)."""
        self.first_stage_model = None
        if self.vtokens:
            self.first_stage_vocab_size = 16384                                    # reference :192
            return
        ckpt = config.params.ckpt_path if (_cfg_has(config, "params") and _cfg_has(config.params, "ckpt_path")) else None
        if ckpt is not None:
            from .vqgan import load_vqgan
            fs = load_vqgan(ckpt)
            for p in fs.parameters():
                p.requires_grad = False
            fs.eval()
            fs.train = disabled_train.__get__(fs)                                  # reference :188
            self.first_stage_model = fs
            self.first_stage_vocab_size = fs.codebook.n_codes                      # reference :189
        else:
            self.first_stage_vocab_size = self.config.first_stage_vocab_size if _cfg_has(self.config, "first_stage_vocab_size") else self.config.vocab_size

    def init_cond_stage_from_ckpt(self, args):
        """reference :204-214 (unconditional only)"""
        if self.be_unconditional:
            self.cond_stage_key = self.first_stage_key
            self.cond_stage_model = SOSProvider(self.sos_token)
            self.cond_stage_vocab_size = 0
        else:
            raise ValueError('conditional model %s is not implementated' % self.cond_stage_key)

    # ---- engine plumbing ------------------------------------------------------------------------------
    def _param_dict(self):
        """the transformer's own parameters (the frozen first stage, when present, is a separate engine: mebt_amd/vqgan.py)"""
        return {k: v for k, v in self.named_parameters() if not k.startswith("first_stage_model.")}

    def _ensure_native(self):
        dev = self.tok_emb.weight.device
        if dev.type != "cuda":
            raise RuntimeError("mebt_amd runs on MI355X only: move the model to the GPU (`model.cuda()`); "
                               "there is no CPU path in the product")
        nm = self._native
        if nm is not None and nm.device == dev and self.tok_emb.weight.data_ptr() == self._tok_ptr:
            return nm
        cfg = self.config
        modes = [b.mode for b in self.transformer.blocks]
        nm = NativeModel(cfg.n_layer, cfg.n_head, cfg.n_embd, self.transformer.head.weight.shape[0], cfg.sos_emb,
                         cfg.block_size, modes, dtype=self.compute_dtype, label_smoothing=self.label_smoothing,
                         embd_pdrop=cfg.embd_pdrop, resid_pdrop=cfg.resid_pdrop, attn_pdrop=cfg.attn_pdrop)
        nm.allocate(dev, with_grads=False)
        params = self._param_dict()
        views = nm.views({k: tuple(v.shape) for k, v in params.items()})
        with torch.no_grad():
            for k, p in params.items():
                views[k].copy_(p.data)
                p.data = views[k]           # the Parameter object survives; its storage is now the flat buffer
        Wn, _ = flat_layout(cfg.n_layer, has_sos=cfg.sos_emb > 0)
        nm.weight_params = [params[k] for k in Wn]
        nm.sync_lowp(force=True)
        self._native, self._tok_ptr = nm, self.tok_emb.weight.data_ptr()
        return nm

    def _attach_grads(self):
        nm = self._native
        if getattr(self, "_grads_attached", None) is nm.gW:
            return
        gv = nm.views({k: tuple(v.shape) for k, v in self._param_dict().items()}, grads=True)
        for k, p in self._param_dict().items():
            p.grad = gv[k]
        self._grads_attached = nm.gW

    def _next_seed(self):
        self._seed_ctr += 1
        return (self.global_step << 20) ^ self._seed_ctr

    def _bucket_hook(self, stage, hi, lo):
        if getattr(self._native, "gWb", None) is not None:
            # a sharded TrainLoop bound the bf16 wire-gradient buffer: the Linear weight gradients of this backward went to gWb
            # only, which the autograd path (fp32 .grad views, bucket-wise all-reduce) never reads
            raise RuntimeError("this model's weight gradients are bound to the bf16 wire buffer of a sharded data-parallel "
                               "TrainLoop; drive it with TrainLoop.step(), or build the loop with wire='fp32' / MEBT_DP_WIRE_GRADS=0")
        if self._reducer is not None:
            self._reducer.bucket_ready(self._native, stage, hi, lo)

    def _trigger(self, dev):
        return torch.zeros((), device=dev, requires_grad=True)

    def _gpt_forward_embedded(self, sos_emb, contexts, targets):
        nm = self._ensure_native()
        if torch.is_grad_enabled() and (self.transformer.training or any(t.requires_grad for t in (sos_emb, contexts, targets))):
            return _GptFn.apply(self, sos_emb, contexts, targets)
        return nm.gpt_forward(sos_emb, contexts, targets)

    def state_dict(self, *args, **kwargs):
        red = self._reducer
        if red is not None and getattr(red, "master_stale", False):
            raise RuntimeError("data-parallel sharded optimizer: the fp32 master weights are complete only on their owning "
                               "ranks; call TrainLoop.consolidate() on ALL ranks before state_dict() / saving a checkpoint")
        if red is not None and getattr(red, "active", False):
            red.finish()                        # gathers / optimizer work of the last step still in flight on other streams
        return super().state_dict(*args, **kwargs)

    # ---- forward ----------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_to_c(self, c):
        """reference :696-701: the conditioning stage's (quantised, indices) pair; the only conditioning stage that exists is the
        unconditional SOSProvider (:204-212), whose pair is the sos token twice, [B, 1]"""
        quant_c, indices = self.cond_stage_model.encode(c, include_embeddings=True)
        if len(indices.shape) > 2:
            indices = indices.view(c.shape[0], -1)
        return quant_c, indices

    @torch.no_grad()
    def encode_to_z(self, x):
        """reference :683-694: token grids pass through (`vtokens`); pixel videos [B,C,T,H,W] go through the first stage's
        encoder + codebook search -> (quantised embeddings [B,t,h,w,c], token ids [B, t*h*w])"""
        if x.dtype == torch.long:
            return x, x.reshape(x.shape[0], -1)
        if self.first_stage_model is None:
            raise NotImplementedError("pixel-space input needs the 3D-VQGAN first stage: build the model with vtokens: False and "
                                      "a first-stage checkpoint (or attach model.first_stage_model = mebt_amd.vqgan.VQGAN(args)), "
                                      "or pass int64 token grids [B,T,H,W]")
        emb, targets = self.first_stage_model.encode(x, include_embeddings=True)
        if self.sample_every_n_latent_frames > 0:
            emb = emb[:, :, ::self.sample_every_n_latent_frames]
            targets = targets[:, ::self.sample_every_n_latent_frames]
        return emb.permute(0, 2, 3, 4, 1), targets.reshape(targets.shape[0], -1).contiguous()

    def _draw_t(self, debug):
        """reference :225-241: ONE scalar t per step from the python RNG (or the beta schedule)"""
        if self.beta and (self.training or debug):
            if self.global_step > self.beta_iter:
                alpha, beta = 1., 1.
            else:
                a_, b_ = self.beta_params
                alpha = a_ - (a_ - 1.) * (self.global_step / self.beta_iter)
                beta = b_ - (b_ - 1.) * (self.global_step / self.beta_iter)
            return float(torch.distributions.beta.Beta(alpha, beta).sample())
        t = random.random()
        if self.training or debug:
            t = self.range[0] + t * (self.range[1] - self.range[0])
        return t

    def forward(self, x, c, t=None, indices=None, vid_t=None, debug=False):
        """one step to produce the logits (reference :216-286) -> (logits, z_targets, NT_weight, seq_len)"""
        assert indices is not None
        _, x_ids = self.encode_to_z(x)
        nm = self._ensure_native()
        if t is None:
            t = self._draw_t(debug)
        if vid_t is None:
            prior_t = self.t_prior(self.t_lengths, self.global_step)
            vid_t = self.t_lengths
        else:
            assert len(vid_t) == 1
            prior_t = np.ones_like(vid_t, dtype=float)
        ci, ti, seq_len = self.mask_sampler.divide_indices(indices, torch.tensor(float(t)), vid_t, prior_t, debug)
        ci, ti = ci.contiguous(), ti.contiguous()
        z_targets = torch.gather(x_ids, 1, ti)
        NT_weight = float(seq_len - ci.shape[1])
        if torch.is_grad_enabled() and self.training:
            logits = _LogitsFn.apply(self._trigger(x_ids.device), self, x_ids, ci, ti)
        else:   # no autograd: optionally keep the activations so the fused loss kernel can run
            logits = nm.forward(x_ids, ci, ti, training=getattr(self, "_keep_for_loss", False), dropout=False)
        return logits, z_targets, NT_weight, seq_len

    def reconstruct_mask(self, x_indices, context_indices, target_indices, debug=False):
        """inference forward on caller-supplied index sets (reference :288-324) -> (logits, None)"""
        B = x_indices.shape[0]
        x_ids = x_indices.reshape(B, -1)
        nm = self._ensure_native()
        return nm.forward(x_ids, context_indices, target_indices, training=False), None

    def _kv_scope(self, B, N):
        """context manager: the outermost sampling loop of a bf16 model with latent_enc blocks owns a key / value cache (_KvSession);
        MEBT_KV_CACHE=0 switches it off (every forward then re-projects every context position, as the reference does)"""
        import contextlib

        @contextlib.contextmanager
        def scope():
            nm = self._ensure_native()
            own = (getattr(self, "_kv", None) is None and nm.dtype == "bf16" and os.environ.get("MEBT_KV_CACHE", "1") != "0"
                   and nm.lib.mebt_kvcache_bytes(nm.h, 1, 1) > 0 and not getattr(nm, "has_maskgit", False))
            if own:
                self._kv = _KvSession(nm, B, N)
            try:
                yield
            finally:
                if own:
                    self._kv_last = (self._kv.rows_projected, self._kv.rows_uncached)
                    self._kv = None
        return scope()

    def _sampling_logits(self, x_indices, context_indices, target_indices, top_p=None, temperature=1.0, dirty=None):
        """the forward of the sampling loops (sample / draft / revise).  A bf16 model hands its logits to the draw kernel in bf16 —
        the fp32 [B, NT, V] tensor was written once and read once just to be sampled from (2.1 GB each way at block 8192) — unless
        the draw needs the generic kernel (top-p, another vocabulary), MEBT_SAMPLE_BF16_LOGITS=0, or the temperature is below 0.5:
        a bf16 logit carries 8 significant bits, so several candidates can round to the same maximum and a (near-)greedy draw
        (`draft_t = 0.0` of the shipped scripts) would pick among them by its noise — at temperature >= 0.5 the rounding (<= 0.4 %
        of a logit) is far below the draw's own randomness.  The fp32 engine (the reference's arithmetic) and the public
        `reconstruct_mask` keep fp32 logits."""
        nm = self._ensure_native()
        lp = (nm.dtype == "bf16" and not top_p and nm.vocab == 16384 and float(temperature) >= 0.5
              and os.environ.get("MEBT_SAMPLE_BF16_LOGITS", "1") != "0" and os.environ.get("MEBT_SAMPLE_FAST", "1") != "0")
        B = x_indices.shape[0]
        kv = getattr(self, "_kv", None)
        if kv is not None:          # the loop's key / value cache: only `dirty` (default: the previous forward's targets) is re-projected
            return kv.forward(x_indices.reshape(B, -1), context_indices.reshape(B, -1), target_indices.reshape(B, -1), dirty, lp)
        if lp:
            return nm.forward(x_indices.reshape(B, -1), context_indices, target_indices, training=False, logits_bf16=True)
        return self.reconstruct_mask(x_indices, context_indices, target_indices)[0]

    def top_k_logits(self, logits, k):
        return top_k_logits(logits, k)

    # ---- loss / training hooks -------------------------------------------------------------------------
    def get_input(self, key, batch):
        return batch[key]

    def get_xc(self, batch, N=None):
        x = self.get_input(self.first_stage_key, batch)
        c = self.get_input(self.cond_stage_key, batch)
        if N is not None:
            x, c = x[:N], c[:N]
        return x, c

    def shared_step(self, batch, batch_idx):
        """reference :717-732 with the loss + top-1/top-5 fused into one pass over the logits
        (mebt_loss) and its backward fused with the CE gradient (mebt_backward_head)."""
        x, c = self.get_xc(batch)
        indices = self.get_input('indices', batch)
        self._keep_for_loss = True
        try:
            logits, target, NT_weight, seq_len = self(x, c, indices=indices)
        finally:
            self._keep_for_loss = False
        ratio = NT_weight / float(seq_len)
        B = logits.shape[0]
        scale = 1.0 / (B * seq_len * ratio ** self.config.avg_loss)
        if logits.requires_grad:
            loss, stats = _LossFn.apply(self._trigger(logits.device), self, logits.detach(), scale)
        else:
            stats = self._native.loss_stats(logits)
            loss = (stats[0] * scale).to(torch.float32)
        n = stats[3]
        acc1 = (stats[1] * (100.0 / n)).to(torch.float32)
        acc5 = (stats[2] * (100.0 / n)).to(torch.float32)
        return acc1, acc5, loss, ratio

    def on_train_epoch_start(self):
        epo = self.current_epoch           # reference :332-334 (reads the epoch, changes nothing)
        return None

    def on_validation_epoch_start(self):
        """Every `vis_epoch` epochs: four clips sampled from an all-masked grid (32 MaskGIT steps, cosine schedule, context
        temperature 6.0), decoded by the first stage and handed to the logger as `[4, T, C, H, W]` in [0, 1] (reference
        :336-351).  The reference dereferences `self.first_stage_model` / `self.logger` unconditionally and so dies here in a
        `vtokens` run without a first stage; this mirror skips the visualisation in that case (warning once)."""
        if (self.current_epoch + 1) % self.vis_epoch != 0:
            return
        if self.first_stage_model is None or self.logger is None:
            if not getattr(self, "_vis_warned", False):
                import warnings
                warnings.warn("on_validation_epoch_start: no first stage / logger attached, sample visualisation skipped")
                self._vis_warned = True
            return
        import copy
        orig_schedule = copy.deepcopy(self.mask_sampler.schedule)
        self.mask_sampler.schedule = 'cosine'
        was_training = self.transformer.training
        self.transformer.eval()
        try:
            shape = (4, *self.mask_sampler.shape)
            x = torch.zeros(shape, dtype=torch.long, device=self.device)
            with torch.no_grad():
                x = self.sample(x, None, 1.0, None, None, 32, None, None, context_temperature=6.0, skips=False)[0]
                code_map = x.reshape(*shape)
                img_x = [self.first_stage_model.decode(code_map[i:i + 1]) for i in range(code_map.shape[0])]
            img_x = torch.cat(img_x, 0).clamp(-0.5, 0.5) + 0.5
            img_x = img_x.permute(0, 2, 1, 3, 4)
            self.logger.experiment.add_video('sample', img_x, self.current_epoch, fps=20)
            self.logger.experiment.flush()
        finally:
            self.mask_sampler.schedule = orig_schedule
            self.transformer.train(was_training)

    def training_step(self, batch, batch_idx):
        acc1, acc5, loss, ratio = self.shared_step(batch, batch_idx)
        self.log("train/loss", loss, prog_bar=True, logger=True, on_step=True, on_epoch=True, sync_dist=True)
        self.log('train/acc1', acc1, prog_bar=True, logger=True, on_step=True, on_epoch=True, sync_dist=True)
        self.log('train/acc5', acc5, prog_bar=True, logger=True, on_step=True, on_epoch=True, sync_dist=True)
        return loss

    def validation_step(self, batch, batch_idx):
        with torch.no_grad():
            acc1, acc5, loss, ratio = self.shared_step(batch, batch_idx)
        self.log("val/loss", loss, prog_bar=True, logger=True, on_step=True, on_epoch=True, sync_dist=True)
        self.log('val/acc1', acc1, prog_bar=True, logger=True, on_step=True, on_epoch=True, sync_dist=True)
        self.log('val/acc5', acc5, prog_bar=True, logger=True, on_step=True, on_epoch=True, sync_dist=True)
        return loss

    def configure_optimizers(self):
        """The 4 AdamW groups of reference :749-798: Linear weights of `transformer` (decayed),
        tok/mask/sos embeddings, biases + LayerNorm, pos_emb (all undecayed); betas (0.9, 0.95)."""
        decay, no_decay = set(), set()
        for mn, m in self.transformer.named_modules():
            for pn, _ in m.named_parameters(recurse=False):
                fpn = f"{mn}.{pn}" if mn else pn
                if pn.endswith("bias") or isinstance(m, (nn.LayerNorm, nn.Embedding)):
                    no_decay.add(fpn)
                elif isinstance(m, nn.Linear):
                    decay.add(fpn)
        pd = dict(self.transformer.named_parameters())
        assert not (decay & no_decay) and not (pd.keys() - (decay | no_decay))
        emb = [p for n, p in self.named_parameters() if "_emb" in n and n != "pos_emb"]
        pos = [p for n, p in self.named_parameters() if "pos_emb" in n]
        groups = [
            {"params": [pd[n] for n in sorted(decay)], "weight_decay": self.weight_decay},
            {"params": emb, "weight_decay": 0.0},
            {"params": [pd[n] for n in sorted(no_decay)], "weight_decay": 0.0},
            {"params": pos, "weight_decay": 0.0},
        ]
        return MebtAdamW(self, groups, lr=self.learning_rate, betas=(0.9, 0.95))

    def lr_scale(self):
        """manual warm-up / cosine of reference :665-678"""
        step = self.trainer.global_step
        if step < self.warmup_steps:
            return min(1., float(step + 1) / self.warmup_steps)
        if self.cosine_lr:
            rad = float(step - self.warmup_steps) / float(self.trainer.max_steps - self.warmup_steps)
            assert rad >= 0
            return 0.5 * (1 + np.cos(rad * np.pi))
        return 1.

    def optimizer_step(self, epoch_nb=0, batch_nb=0, optimizer=None, optimizer_i=0, opt_closure=None, on_tpu=False,
                       using_native_amp=False, using_lbfgs=False):
        scale = self.lr_scale()
        if self.trainer.global_step < self.warmup_steps or self.cosine_lr:
            for pg in optimizer.param_groups:
                pg['lr'] = self.learning_rate * scale
        self.log("learning_rate", self.learning_rate * scale, logger=True, on_step=True, sync_dist=True)
        optimizer.step(closure=opt_closure)

    # ---- sampling -----------------------------------------------------------------------------------------
    def _noise(self, kind, shape, device):
        if self.noise_hook is not None:
            return self.noise_hook(kind, tuple(shape)).to(device, torch.float32)
        return torch.empty(shape, device=device, dtype=torch.float32).exponential_()

    def _sample_tokens(self, logits, temperature, top_k, top_p, want_probs=False, probs_map=None, target_indices=None):
        """sample_from_logits (:843-889) + score gather (:409) in one kernel -> (ids, scores, probs?).  `probs_map` [B, N, V] with
        `target_indices` [B, NT]: the probabilities go straight to rows target_indices of the map (the `scatter_` of debug=True,
        :426-436) instead of through a [B, NT, V] tensor; returns probs = None then."""
        B, NT, V = logits.shape
        noise = None if self.noise_hook is None else self._noise("exp", (B, NT, V), logits.device)
        # production (no hook): Exp(1) generated inside the kernel, seeded from torch's default generator
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if noise is None else None
        fast = os.environ.get("MEBT_SAMPLE_FAST", "1") != "0"       # 0: the round-3 kernel for every shape (it has no scattered-map form)
        # Production draw (no injected noise), register kernel: inverse CDF from one uniform per row — a sample of the same categorical
        # distribution the reference's arg-max p / q, q ~ Exp(1), draws (:826-841), without its per-element hash / log / reciprocal
        # (include/mebt_hip.h: mebt_op_sample_lp, draw = 1).  MEBT_SAMPLE_ICDF=0: arg-max p / q with in-kernel Exp(1) noise.  With
        # injected noise (tests, parity) the draw is always the reference's arithmetic.
        icdf = noise is None and not top_p and V == 16384 and fast and not want_probs and os.environ.get("MEBT_SAMPLE_ICDF", "1") != "0"
        if logits.dtype == torch.bfloat16 or icdf:         # bf16: the head's bf16 output (_sampling_logits)
            assert not top_p and V == 16384 and fast
            bf = logits.dtype == torch.bfloat16
            lg = logits.contiguous() if bf else logits.to(torch.float32).contiguous()
            ids = torch.empty(B, NT, dtype=torch.long, device=lg.device)
            score = torch.empty(B, NT, dtype=torch.float32, device=lg.device)
            ti = target_indices.contiguous() if probs_map is not None else None
            nz = None if noise is None else noise.to(torch.float32).contiguous()
            _lib.check(_lib.load().mebt_op_sample_lp(_lib.ptr(lg), 1 if bf else 0, _lib.ptr(nz), int(seed or 0), float(temperature), int(top_k or 0),
                                                     _lib.ptr(ids), _lib.ptr(score), _lib.ptr(probs_map), _lib.ptr(ti), B,
                                                     probs_map.shape[1] if probs_map is not None else NT, NT, V, 1 if icdf else 0,
                                                     _lib.cur_stream()))
            return ids, score, None
        if probs_map is not None and not top_p and V == 16384 and fast:
            lg = logits.to(torch.float32).contiguous()
            ids = torch.empty(B, NT, dtype=torch.long, device=lg.device)
            score = torch.empty(B, NT, dtype=torch.float32, device=lg.device)
            ti = target_indices.contiguous()
            nz = None if noise is None else noise.to(torch.float32).contiguous()
            _lib.check(_lib.load().mebt_op_sample_scatter(_lib.ptr(lg), _lib.ptr(nz), int(seed or 0), float(temperature), int(top_k or 0),
                                                          _lib.ptr(ids), _lib.ptr(score), _lib.ptr(probs_map), _lib.ptr(ti), B, probs_map.shape[1],
                                                          NT, V, _lib.cur_stream()))
            return ids, score, None
        ids, score, probs = sample_from_logits_scored(logits, temperature, top_k, top_p, noise, want_probs or probs_map is not None, seed=seed)
        if probs_map is not None:
            probs_map.scatter_(1, target_indices.unsqueeze(-1).expand(-1, -1, V), probs)
            probs = None
        return ids, score, probs

    @staticmethod
    def _scatter(partial, target_indices, ids):
        """x[b, ti[b,j]] = ids[b,j]  — replaces the two sparse_coo -> dense -> where of :413-439"""
        B, N = partial.shape
        out = partial.clone()
        ti = target_indices.contiguous()
        ids = ids.contiguous()
        _lib.check(_lib.load().mebt_op_scatter_ids(_lib.ptr(out), _lib.ptr(ti), _lib.ptr(ids), B, N, ti.shape[1],
                                                   _lib.cur_stream()))
        return out

    @torch.no_grad()
    def sample(self, x, c, temperature=1.0, top_k=None, top_p=None, n_steps=8, context_indices=None,
               target_indices=None, strategy='maskgit', context_temperature=4.5, phase_history=None, refine_steps=1,
               forget_pivot=False, skips=[False, False, False], debug=False, ctemp_schedule='linear', edit=False, _chosen_probs=False):
        """MaskGIT-style iterative decoding (reference :353-447).  `_chosen_probs` (with debug=True; this package's drivers only):
        the sixth element of the tuple is a [B, N] map of the probability of the id chosen at each position (-1 where never sampled)
        instead of the [B, N, V] probability map — all `bidirect_sample` reads from that map is its value at the chosen code."""
        B = x.shape[0]
        N = int(np.prod(x.shape[1:]))
        edit_N = target_indices.shape[1] if edit else N
        x = x.reshape(B, N)
        assert not self.transformer.training
        if strategy not in ('maskgit', 'random', 'mlm', 'bootstrap'):
            return None
        if ctemp_schedule not in CTEMP_SCHEDULES:
            raise ValueError(ctemp_schedule)
        self.mask_sampler.noise_hook = self.noise_hook
        dev = x.device
        if context_indices is None:
            context_indices = torch.empty(B, 0, dtype=torch.long, device=dev)
            target_indices = torch.arange(N, device=dev).repeat(B, 1)
        else:
            context_indices, target_indices = context_indices.clone(), target_indices.clone()
        partial = x
        history, context_history = [], []
        if debug:
            history.append(partial.clone())
            V = self.transformer.head.weight.shape[0]
            partial_probs = -torch.ones((B, N) if _chosen_probs else (B, N, V), device=dev)
        with self._kv_scope(B, N):
            return self._sample_loop(B, N, edit_N, partial, context_indices, target_indices, temperature, top_k, top_p, n_steps, strategy,
                                     context_temperature, ctemp_schedule, debug, history, context_history,
                                     partial_probs if debug else None)

    def _sample_loop(self, B, N, edit_N, partial, context_indices, target_indices, temperature, top_k, top_p, n_steps, strategy,
                     context_temperature, ctemp_schedule, debug, history, context_history, partial_probs):
        nc_done = None                     # context entries whose keys / values the cache already holds (None: nothing yet)
        for t_next in np.linspace(0, 1, n_steps + 1)[1:]:
            tt = torch.full((B,), fill_value=t_next)                        # float32, like reference :398
            n_masked = torch.ceil(self.mask_sampler.schedule_fn(tt) * edit_N)
            if int((n_masked > target_indices.shape[-1]).sum()) == B:       # :401-402
                continue
            # generate_next_mask appends the newly fixed tokens to the context (mask_sampler.py:228-233): only they are new to the cache
            dirty = None if nc_done is None else context_indices[:, nc_done:]
            logits = self._sampling_logits(partial, context_indices, target_indices, top_p, temperature, dirty=dirty)
            nc_done = context_indices.shape[1]
            target_indices = target_indices.view(B, -1)
            chosen_only = debug and partial_probs is not None and partial_probs.dim() == 2
            ids, scores, _ = self._sample_tokens(logits, temperature, top_k, top_p, probs_map=partial_probs if debug and not chosen_only else None,
                                                 target_indices=target_indices)
            if chosen_only:        # the map's value at the chosen id of every current target = the draw's score (:409, :426-436)
                partial_probs.scatter_(1, target_indices, scores)
            partial = self._scatter(partial, target_indices, ids)
            ctemp = context_temperature * CTEMP_SCHEDULES[ctemp_schedule](t_next)   # :440
            if debug:
                history.append(partial.clone())
                context_history.append(context_indices)
            context_indices, target_indices = self.mask_sampler.generate_next_mask(
                context_indices, target_indices, scores, t_next, strategy=strategy, context_temperature=ctemp,
                n_masked_toks=n_masked)
        if debug:
            return partial.view(B, -1), context_indices, target_indices, history, context_history, partial_probs
        return partial.view(B, -1), context_indices, target_indices

    def _gibbs_pass(self, x, masks, temperature, top_k, top_p, debug, kind):
        """one draft (`kind` = 'draft') or revise pass over its list of (context, target) index sets.  Key / value cache: the only
        context positions whose token changed since the previous forward are that forward's targets that are context now — by
        construction the LAST w columns of this context set (mask_sampler.py:334 / :354: revise appends the chunk just re-sampled,
        draft the chunk just fixed), w = the previous chunk's width."""
        partial = x
        B = x.shape[0]
        prev_tgt = None
        with self._kv_scope(B, x.shape[1]):
            for ctx, tgt in zip(*masks):
                ctx, tgt = ctx.contiguous(), tgt.contiguous()
                dirty = None
                if prev_tgt is not None:
                    w = prev_tgt.shape[1] - (tgt.shape[1] if kind == "draft" else 0)
                    dirty = ctx[:, ctx.shape[1] - w:] if 0 < w <= ctx.shape[1] else None
                logits = self._sampling_logits(partial, ctx, tgt, top_p, temperature, dirty=dirty)
                ids, _, _ = self._sample_tokens(logits, temperature, top_k, top_p)
                partial = self._scatter(partial, tgt.reshape(B, -1), ids)
                prev_tgt = tgt
        return partial

    def _full_sets(self, B, N, device, context_indices, target_indices):
        if context_indices is None:
            context_indices = torch.empty(B, 0, dtype=torch.long, device=device)
            target_indices = torch.arange(N, device=device).repeat(B, 1)
        return context_indices, target_indices

    @torch.no_grad()
    def draft(self, x, c, temperature=1.0, top_k=None, top_p=None, n_steps=8, debug=False, context_indices=None,
              target_indices=None):
        """reference :544-586"""
        B, N = x.shape[0], int(np.prod(x.shape[1:]))
        x = x.reshape(B, N)
        self.mask_sampler.noise_hook = self.noise_hook
        ci, ti = self._full_sets(B, N, x.device, context_indices, target_indices)
        masks = self.mask_sampler.create_gibbs_draft_mask(ci, ti, n_steps, x.device)
        assert not self.transformer.training
        return self._gibbs_pass(x, masks, temperature, top_k, top_p, debug, "draft").view(B, -1)

    @torch.no_grad()
    def revise(self, x, c, temperature=1.0, top_k=None, top_p=None, n_steps=8, debug=False, context_indices=None,
               target_indices=None):
        """reference :588-630"""
        B, N = x.shape[0], int(np.prod(x.shape[1:]))
        x = x.reshape(B, N)
        self.mask_sampler.noise_hook = self.noise_hook
        ci, ti = self._full_sets(B, N, x.device, context_indices, target_indices)
        masks = self.mask_sampler.create_gibbs_revise_mask(ci, ti, n_steps, x.device)
        assert not self.transformer.training
        return self._gibbs_pass(x, masks, temperature, top_k, top_p, debug, "revise").view(B, -1)

    @torch.no_grad()
    def draft_and_revise(self, x, c, n_draft=8, draft_t=1.0, draft_k=None, draft_p=None, n_revise=8, revise_t=1.0,
                         revise_k=None, revise_p=None, M=2, skip_draft=False, debug=False, context_indices=None,
                         target_indices=None, edit=False):
        """reference :632-663"""
        B, N = x.shape[0], int(np.prod(x.shape[1:]))
        x = x.reshape(B, N)
        assert not self.transformer.training
        with self._kv_scope(B, N):          # one key / value cache for the draft and every revise pass (the first forward of a pass
            if not skip_draft:             # re-projects the previous pass's last chunk: _KvSession's default `dirty`)
                x = self.draft(x, c, draft_t, draft_k, draft_p, n_draft, debug, context_indices, target_indices)
            if edit:
                context_indices = target_indices = None
            for _ in range(M):
                x = self.revise(x, c, revise_t, revise_k, revise_p, n_revise, debug, context_indices, target_indices)
        return x.view(B, -1)


# ---- module-level sampler helpers (reference :826-910) -----------------------------------------------------
def sample_from_logits_scored(logits, temperature, top_k, top_p, noise, want_probs=False, seed=None):
    """ids = argmax(p/noise), scores = p[ids] (and optionally p) in ONE kernel; p is the distribution
    after temperature / top-k / top-p (reference :859-874).  noise = None: q ~ Exp(1) is drawn inside the kernel from the
    counter-based generator keyed by `seed` (no [rows, V] noise tensor through HBM)."""
    shape = logits.shape[:-1]
    V = logits.shape[-1]
    lg = logits.to(torch.float32).contiguous().view(-1, V)
    R = lg.shape[0]
    if noise is None:
        ids = torch.empty(R, dtype=torch.long, device=lg.device)
        score = torch.empty(R, dtype=torch.float32, device=lg.device)
        probs = torch.empty(R, V, dtype=torch.float32, device=lg.device) if want_probs else None
        _lib.check(_lib.load().mebt_op_sample_seeded(_lib.ptr(lg), int(seed or 0), float(temperature), int(top_k or 0), float(top_p or 0.0),
                                                     _lib.ptr(ids), _lib.ptr(score), _lib.ptr(probs), R, V, _lib.cur_stream()))
        return ids.view(shape), score.view(shape), (probs.view(*shape, V) if want_probs else None)
    nz = noise.to(torch.float32).contiguous().view(-1, V)
    ids = torch.empty(R, dtype=torch.long, device=lg.device)
    score = torch.empty(R, dtype=torch.float32, device=lg.device)
    probs = torch.empty(R, V, dtype=torch.float32, device=lg.device) if want_probs else None
    _lib.check(_lib.load().mebt_op_sample(_lib.ptr(lg), _lib.ptr(nz), float(temperature), int(top_k or 0),
                                          float(top_p or 0.0), _lib.ptr(ids), _lib.ptr(score), _lib.ptr(probs), R, V,
                                          _lib.cur_stream()))
    return ids.view(shape), score.view(shape), (probs.view(*shape, V) if want_probs else None)


def sample_from_logits(logits, temperature=1.0, top_k=None, top_p=None, return_probs=False, noise=None):
    """reference :843-889 (same signature + an optional explicit `noise` tensor of Exp(1) draws)"""
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if noise is None else None
    ids, _, probs = sample_from_logits_scored(logits, temperature, top_k, top_p, noise, want_probs=return_probs, seed=seed)
    return (ids, probs) if return_probs else ids


def top_k_logits(logits, k):
    """reference :891-895 (also the `Net2NetTransformer.top_k_logits` method): everything below the k-th largest logit of a
    row becomes -inf, ties with the k-th value are kept.  The k-th value comes from the sampler kernel's radix select
    (csrc/sampler.hip, `mebt_op_topk_threshold`); kept entries keep their value however small (a mask derived from
    the filtered softmax also dropped kept entries whose probability underflows)."""
    if not logits.is_cuda:
        raise RuntimeError("mebt_amd has no CPU path: top_k_logits needs a tensor on the MI355X (cuda) device")
    V = logits.shape[-1]
    if k >= V:
        return logits.clone()
    lg = logits.to(torch.float32).contiguous().view(-1, V)
    R = lg.shape[0]
    thr = torch.empty(R, dtype=torch.float32, device=lg.device)
    scratch = torch.empty(R, dtype=torch.long, device=lg.device)
    _lib.check(_lib.load().mebt_op_topk_threshold(_lib.ptr(lg), int(k), _lib.ptr(thr), _lib.ptr(scratch), R, V, _lib.cur_stream()))
    return logits.masked_fill(logits.to(torch.float32) < thr.view(*logits.shape[:-1], 1), -float("Inf"))


def top_p_probs(probs, p):
    """reference :898-910: nucleus filter of a probability tensor (*, C) -> renormalised probabilities (the smallest prefix of
    the descending order whose mass reaches p is kept).  Runs on the sampler kernel (its top-p stage on log(probs) at
    temperature 1, csrc/sampler.hip): device tensors only."""
    if not probs.is_cuda:
        raise RuntimeError("mebt_amd has no CPU path: top_p_probs needs a tensor on the MI355X (cuda) device")
    logits = torch.log(probs.to(torch.float32).clamp_min(0))
    _, _, out = sample_from_logits_scored(logits, 1.0, None, float(p), None, want_probs=True, seed=0)
    return out.to(probs.dtype)


def gumbel_sort(prob, noise=None):
    """reference :826-841: indices (*, C) in descending order of (prob / sum) / q with q ~ Exp(1), entries of probability 0 last.
    API parity only — the sampling path never sorts (`sample_from_logits` takes the arg-max inside its kernel); `noise` replaces
    the draw for tests."""
    prob = prob / prob.sum(-1, keepdim=True)
    q = torch.empty_like(prob).exponential_() if noise is None else noise.to(prob)
    keyed = (prob / q) * (prob > 0).to(prob.dtype)
    return keyed.sort(dim=-1, descending=True)[1]
