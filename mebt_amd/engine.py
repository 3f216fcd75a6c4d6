"""Host-side owner of the native model: flat parameter / gradient buffers, the workspace and the
calls into libmebt_hip.so.  PyTorch allocates every byte; the library never does (SURVEY.md §8b
ownership).  Parameter tensors of the nn.Module tree become *views* into the two flat buffers so
state_dict / load_state_dict / optimiser param_groups keep working with the reference's names.
"""
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import check, ptr, cur_stream


def layer_param_names(i):
    p = f"transformer.blocks.{i}."
    w = [p + "attn.query.weight", p + "attn.key.weight", p + "attn.value.weight", p + "attn.proj.weight",
         p + "mlp.0.weight", p + "mlp.2.weight"]
    s = [p + "ln1.weight", p + "ln1.bias", p + "ln2.weight", p + "ln2.bias", p + "attn.query.bias",
         p + "attn.key.bias", p + "attn.value.bias", p + "attn.proj.bias", p + "mlp.0.bias", p + "mlp.2.bias"]
    return w, s


def flat_layout(n_layer, has_sos=True):
    """Order of the state-dict tensors inside the W (nn.Linear weights) and P (everything else)
    flat buffers — must match mebt_model_create in csrc/engine.cpp."""
    W, P = [], []
    for i in range(n_layer):
        w, s = layer_param_names(i)
        W += w
        P += s
    W.append("transformer.head.weight")
    P += ["transformer.ln_f.weight", "transformer.ln_f.bias", "mask_emb"]
    if has_sos:
        P.append("sos_emb")
    P += ["pos_emb", "tok_emb.weight"]
    return W, P


class NativeModel:
    """One mebt_model handle bound to flat buffers on one GPU."""

    def __init__(self, n_layer, n_head, n_embd, vocab, n_latent, block_size, modes, dtype="bf16",
                 label_smoothing=0.0, embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0):
        self.lib = _lib.load()
        d = _lib.ModelDesc()
        d.n_layer, d.n_head, d.n_embd, d.vocab = n_layer, n_head, n_embd, vocab
        d.n_latent, d.block_size = n_latent, block_size
        self.dtype = dtype
        d.dtype = {"f32": _lib.F32, "fp32": _lib.F32, "bf16": _lib.BF16}[dtype]
        if len(modes) > _lib.MEBT_MAX_LAYERS:
            raise ValueError("too many layers")
        for i, m in enumerate(modes):
            if m not in _lib.MODE_IDS:
                raise NotImplementedError(f"block mode {m!r} is not supported by the HIP engine (supported: {sorted(_lib.MODE_IDS)})")
            d.modes[i] = _lib.MODE_IDS[m]
        d.label_smoothing = float(label_smoothing)
        d.embd_pdrop, d.resid_pdrop, d.attn_pdrop = float(embd_pdrop), float(resid_pdrop), float(attn_pdrop)
        self.desc = d
        h = C.c_void_p()
        check(self.lib.mebt_model_create(C.byref(d), C.byref(h)))
        self.h = h
        nw, np_ = C.c_int64(), C.c_int64()
        check(self.lib.mebt_model_param_counts(self.h, C.byref(nw), C.byref(np_)))
        self.n_w, self.n_p = nw.value, np_.value
        if os.environ.get("MEBT_GROUPED_STAGES"):
            self.lib.mebt_debug_grouped_stages(int(os.environ["MEBT_GROUPED_STAGES"]))
        # second stream for the gradient leaves of backward: off by default (neutral-to-slower since every launch fills
        # the chip: 11.3 vs 11.6 ms per Sky-16f step), MEBT_SIDE_STREAM=1 turns it on
        self.side_stream = os.environ.get("MEBT_SIDE_STREAM", "0") == "1"
        self.lib.mebt_debug_side_stream(self.h, 1 if self.side_stream else 0)
        self._side = None
        if self.side_stream and os.environ.get("MEBT_SIDE_STREAM_PROBED", "1") == "1" and torch.cuda.is_available():
            from .parallel import pick_concurrent_stream            # a stream that really runs beside the compute stream
            self._side = pick_concurrent_stream(torch.cuda.current_stream())
            self.lib.mebt_debug_set_side_stream(self.h, self._side.cuda_stream)
        self.n_layer, self.n_embd, self.vocab, self.n_latent = n_layer, n_embd, vocab, n_latent
        self.has_maskgit = any(m == "maskgit" for m in modes)      # such blocks rewrite the contexts: no key / value cache
        self.W = self.P = self.gW = self.gP = self.Wlp = None
        self.gWb = None             # bf16 wire-format weight gradients (data-parallel sharded path)
        self.ws = None
        self.ws_key = None
        self._w_version = None
        self.weight_params = []     # the nn.Parameters whose storage is a view of W (set by the module that owns them)
        self.generation = 0         # bumped by every training-mode forward: a backward must match the forward it belongs to
        self.device = None
        self.adam = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.mebt_model_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- buffers ---------------------------------------------------------------------------------
    def allocate(self, device, with_grads=True):
        self.device = torch.device(device)
        kw = dict(device=self.device, dtype=torch.float32)
        self.W = torch.zeros(self.n_w, **kw)
        self.P = torch.zeros(self.n_p, **kw)
        self.Wlp = torch.zeros(self.n_w, device=self.device, dtype=torch.bfloat16) if self.dtype == "bf16" else None
        if with_grads:
            self.gW = torch.zeros(self.n_w, **kw)
            self.gP = torch.zeros(self.n_p, **kw)
        self.bind()

    def ensure_grads(self):
        if self.gW is None:
            self.gW = torch.zeros_like(self.W)
            self.gP = torch.zeros_like(self.P)
            self.bind()

    def bind(self):
        check(self.lib.mebt_model_bind(self.h, ptr(self.W), ptr(self.Wlp), ptr(self.gW), ptr(self.P), ptr(self.gP)))

    def sync_lowp(self, force=False):
        """Refresh the bf16 weight mirror if the fp32 master was modified through torch.  In-place ops bump the version
        counter of the tensor they are applied to: the flat buffer itself, or one of the nn.Parameters that view it
        (`load_state_dict`, `p.mul_()`, a stock torch optimizer) — each Parameter has its own counter, so all of them
        are watched.  Our own AdamW kernels update W and the mirror together and bump nothing."""
        if self.Wlp is None:
            return
        v = (self.W._version, sum(p._version for p in self.weight_params))
        if force or v != self._w_version:
            check(self.lib.mebt_model_sync_lowp(self.h, cur_stream()))
            self._w_version = v

    def workspace(self, B, NC, NT, training):
        need = self.lib.mebt_workspace_bytes(self.h, B, NC, NT, int(training))
        if need < 0:
            raise _lib.MebtError("bad workspace query")
        if self.ws is None or self.ws.numel() < need or self.ws.device != self.device:
            self.ws = torch.empty(int(need * 1.05) + 4096, dtype=torch.uint8, device=self.device)
        return self.ws

    # ---- compute -----------------------------------------------------------------------------------
    def forward(self, x_ids, ci, ti, training=False, logits=None, dropout_seed=0, dropout=True, logits_bf16=False):
        """x_ids [B,N] i64, ci [B,NC], ti [B,NT] -> logits [B,NT,V] fp32.  training keeps the
        activations for backward / the fused loss; dropout (only with training) enables the masks.
        logits_bf16 (inference of a bf16 model): the head stores bf16 logits — for the sampling loops, whose draw kernel reads them."""
        assert x_ids.dtype == torch.long and ti.dtype == torch.long
        x_ids, ti = x_ids.contiguous(), ti.contiguous()
        ci = ci.contiguous() if ci is not None else None
        B, N = x_ids.shape
        NC = 0 if ci is None else ci.shape[1]
        NT = ti.shape[1]
        self.sync_lowp()
        ws = self.workspace(B, NC, NT, training)
        logits_bf16 = bool(logits_bf16) and not training and self.dtype == "bf16"
        if logits is None:
            logits = torch.empty(B, NT, self.vocab, device=self.device, dtype=torch.bfloat16 if logits_bf16 else torch.float32)
        check(self.lib.mebt_forward(self.h, ptr(ws), ws.numel(), B, N, NC, NT, ptr(x_ids),
                                    ptr(ci) if NC > 0 else None, ptr(ti), ptr(logits),
                                    (1 | (2 if dropout else 0)) if training else (4 if logits_bf16 else 0),
                                    int(dropout_seed), cur_stream()))
        self._keep = (x_ids, ci, ti, logits)   # the native context holds raw pointers to these
        if training:
            self.generation += 1
        return logits

    def new_kv_cache(self, B, N):
        """the key / value cache of `forward_cached` for B samples of N positions (include/mebt_hip.h: mebt_forward_kvcache)"""
        nbytes = self.lib.mebt_kvcache_bytes(self.h, int(B), int(N))
        if nbytes <= 0:
            return None
        return torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)

    def forward_cached(self, x_ids, ci, ti, cache, dirty, logits_bf16=False):
        """inference forward with the latent_enc blocks' keys / values read from `cache` at `ci`; the rows of `dirty` [B, ND]
        (None: every position of `ci`) are re-projected first.  bf16 models only."""
        assert self.dtype == "bf16" and x_ids.dtype == torch.long
        x_ids, ti, ci = x_ids.contiguous(), ti.contiguous(), ci.contiguous()
        dirty = ci if dirty is None else dirty.contiguous()
        B, N = x_ids.shape
        NC, NT, ND = ci.shape[1], ti.shape[1], dirty.shape[1]
        self.sync_lowp()
        ws = self.workspace(B, NC, NT, False)
        logits = torch.empty(B, NT, self.vocab, device=self.device, dtype=torch.bfloat16 if logits_bf16 else torch.float32)
        check(self.lib.mebt_forward_kvcache(self.h, ptr(ws), ws.numel(), B, N, NC, NT, ptr(x_ids), ptr(ci) if NC > 0 else None, ptr(ti),
                                            ptr(logits), 4 if logits_bf16 else 0, ptr(cache), ptr(dirty) if ND > 0 else None, ND, cur_stream()))
        self._keep = (x_ids, ci, ti, logits, dirty)
        return logits

    def gpt_forward(self, sos, contexts, targets):
        """GPT.forward boundary: fp32 embeddings in, logits out (inference)."""
        sos, contexts, targets = (t.to(self.device, torch.float32).contiguous() for t in (sos, contexts, targets))
        B, NC, NT = sos.shape[0], contexts.shape[1], targets.shape[1]
        self.sync_lowp()
        ws = self.workspace(B, NC, NT, False)
        logits = torch.empty(B, NT, self.vocab, device=self.device, dtype=torch.float32)
        check(self.lib.mebt_gpt_forward(self.h, ptr(ws), ws.numel(), B, NC, NT, ptr(sos), ptr(contexts) if NC > 0 else None,
                                        ptr(targets), ptr(logits), cur_stream()))
        return logits

    def gpt_forward_train(self, sos, contexts, targets, dropout_seed=0, dropout=True):
        """GPT.forward boundary in training mode (activations kept for gpt_backward)"""
        sos, contexts, targets = (t.to(self.device, torch.float32).contiguous() for t in (sos, contexts, targets))
        B, NC, NT = sos.shape[0], contexts.shape[1], targets.shape[1]
        self.sync_lowp()
        ws = self.workspace(B, NC, NT, True)
        logits = torch.empty(B, NT, self.vocab, device=self.device, dtype=torch.float32)
        check(self.lib.mebt_gpt_forward_train(self.h, ptr(ws), ws.numel(), B, NC, NT, ptr(sos), ptr(contexts) if NC > 0 else None,
                                              ptr(targets), ptr(logits), 1 if dropout else 0, int(dropout_seed), cur_stream()))
        self._keep = (sos, contexts, targets, logits)
        self.generation += 1
        return logits

    def gpt_backward(self, dlogits, shapes):
        """-> (d_sos, d_contexts, d_targets) fp32; parameter gradients of blocks / ln_f / head land in gW, gP"""
        self.ensure_grads()
        dlogits = dlogits.to(torch.float32).contiguous()
        outs = [torch.empty(s, device=self.device, dtype=torch.float32) for s in shapes]
        check(self.lib.mebt_gpt_backward(self.h, ptr(self.ws), ptr(dlogits), *[ptr(o) if o.numel() else None for o in outs], cur_stream()))
        return outs

    def loss_stats(self, logits, grad_scale=None):
        """device tensor [4] float64: CE sum, #top-1, #top-5, #rows (of the last training forward).  `grad_scale`: the loss
        scale the following backward will use — the cross-entropy gradient is then produced by the same pass over the logits
        (mebt_loss_with_grad) and `backward(logits, grad_scale)` skips its own cross-entropy backward."""
        out = torch.empty(4, device=self.device, dtype=torch.float64)
        if grad_scale is None:
            check(self.lib.mebt_loss(self.h, ptr(self.ws), ptr(logits), ptr(out), cur_stream()))
        else:
            check(self.lib.mebt_loss_with_grad(self.h, ptr(self.ws), ptr(logits), ptr(out), float(grad_scale), cur_stream()))
        return out

    def backward_head(self, logits, loss_scale, upstream=None):
        self.ensure_grads()
        check(self.lib.mebt_backward_head(self.h, ptr(self.ws), ptr(logits), ptr(upstream), float(loss_scale), cur_stream()))

    def backward_layers(self, hi, lo):
        check(self.lib.mebt_backward_layers(self.h, ptr(self.ws), hi, lo, cur_stream()))

    def backward_embed(self):
        check(self.lib.mebt_backward_embed(self.h, ptr(self.ws), cur_stream()))

    def backward(self, logits, loss_scale, upstream=None, between=None, dlogits=None, bucket_layers=4):
        """Full backward.  `between(stage, hi, lo)` is called after each finished gradient bucket
        ('head', 'layers', 'embed') so a data-parallel reducer can launch its all-reduce.
        `dlogits` (fp32 [B,NT,V]) replaces the fused cross-entropy backward by an explicit upstream."""
        if dlogits is not None:
            self.ensure_grads()
            dlogits = dlogits.contiguous()
            check(self.lib.mebt_backward_head_dlogits(self.h, ptr(self.ws), ptr(dlogits), cur_stream()))
        else:
            self.backward_head(logits, loss_scale, upstream)
        if between:
            between("head", None, None)
        sizes = list(bucket_layers) if isinstance(bucket_layers, (list, tuple)) else None     # per-bucket layer counts, top down
        hi, nb = self.n_layer - 1, 0
        while hi >= 0:
            step = max(1, int(sizes[min(nb, len(sizes) - 1)] if sizes else bucket_layers))
            nb += 1
            lo = max(0, hi - step + 1)
            self.backward_layers(hi, lo)
            if between:
                between("layers", hi, lo)
            hi = lo - 1
        self.backward_embed()
        if between:
            between("embed", None, None)

    def _adam_state(self):
        if self.adam is None:
            self.adam = [torch.zeros_like(self.W), torch.zeros_like(self.W), torch.zeros_like(self.P), torch.zeros_like(self.P)]
        return self.adam

    def adamw_range(self, kind, hi, lo, lr, weight_decay, step, betas=(0.9, 0.95), eps=1e-8, grad_scale=1.0, stream=None):
        """AdamW on one gradient bucket ('head' | 'layers' | 'embed'), on `stream` (default: current)."""
        mW, vW, mP, vP = self._adam_state()
        k = {"head": 0, "layers": 1, "embed": 2, "all": 3, "rest": 4}[kind]
        check(self.lib.mebt_adamw_range(self.h, ptr(mW), ptr(vW), ptr(mP), ptr(vP), float(lr), float(betas[0]), float(betas[1]),
                                        float(eps), float(weight_decay), int(step), float(grad_scale), k, int(hi or 0), int(lo or 0),
                                        stream if stream is not None else cur_stream()))

    def adamw_slice(self, which, off, n, grad, lr, weight_decay, step, betas=(0.9, 0.95), eps=1e-8, grad_scale=1.0, stream=None, pieces=1):
        """AdamW on elements [off, off+n) of W (which = 0) or P (1) with `grad` (fp32 or bf16, n elements) as gradient; `pieces` > 1:
        `grad` holds that many bf16 copies of the slice's gradient back to back (an all-to-all's receive buffer), summed in fp32"""
        mW, vW, mP, vP = self._adam_state()
        if pieces > 1:
            assert grad.dtype == torch.bfloat16 and grad.numel() == pieces * n
            check(self.lib.mebt_adamw_slice_pieces(self.h, int(which), int(off), int(n), ptr(grad), int(pieces),
                                                   ptr(mW), ptr(vW), ptr(mP), ptr(vP), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                                   float(weight_decay), int(step), float(grad_scale), stream if stream is not None else cur_stream()))
            return
        check(self.lib.mebt_adamw_slice(self.h, int(which), int(off), int(n), ptr(grad), 1 if grad.dtype == torch.bfloat16 else 0,
                                        ptr(mW), ptr(vW), ptr(mP), ptr(vP), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                        float(weight_decay), int(step), float(grad_scale), stream if stream is not None else cur_stream()))

    def cast_bf16(self, src, dst):
        """dst (bf16) = src (fp32), flat, on the current stream (gradient bucket -> wire format)"""
        check(self.lib.mebt_op_cast_bf16(ptr(src), ptr(dst), src.numel(), cur_stream()))

    def set_fused_adamw(self, lr=0.0, weight_decay=0.0, step=0, betas=(0.9, 0.95), eps=1e-8, grad_scale=1.0):
        """Arm (step >= 1) / disarm (step = 0) the optimizer-in-backward for the blocks' Linear weights; after the
        backward call adamw_range('rest', ...) for everything else."""
        mW, vW, _, _ = self._adam_state()
        check(self.lib.mebt_model_set_fused_adamw(self.h, ptr(mW), ptr(vW), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                                  float(weight_decay), int(step), float(grad_scale)))

    def enable_wire_grads(self, on=True):
        """bf16 compute mode: store the Linear weight gradients as bf16 in `gWb` (same layout as gW) instead of fp32 in gW —
        the wire format of the data-parallel reduce-scatter, written once from the fp32 accumulators"""
        if on and self.dtype == "bf16":
            if self.gWb is None:
                self.gWb = torch.zeros(self.n_w, device=self.device, dtype=torch.bfloat16)
            check(self.lib.mebt_model_bind_wire_grads(self.h, ptr(self.gWb)))
        else:
            self.gWb = None
            check(self.lib.mebt_model_bind_wire_grads(self.h, None))

    def set_forward_waits(self, waits):
        """[(layer, torch.cuda.Event)]: the next forward (training or inference) waits, on its stream, for each event before
        it reads the parameters first used by block `layer` (-1: everything outside the Linear weights; n_layer: the head).
        One-shot.  The events are kept alive here until the next call."""
        import ctypes as C
        waits = list(waits)
        n = len(waits)
        layers = (C.c_int32 * max(1, n))(*[int(l) for l, _ in waits])
        evs = (C.c_void_p * max(1, n))(*[int(e.cuda_event) for _, e in waits])
        check(self.lib.mebt_model_set_forward_waits(self.h, n, layers, evs))
        self._fw_events = [e for _, e in waits]

    def set_grad_accumulate(self, on):
        """on: the next backward adds to the gradient buffers (a further micro-batch); off: it overwrites them"""
        check(self.lib.mebt_model_set_grad_accumulate(self.h, 1 if on else 0))

    def adamw_step(self, lr, weight_decay, step, betas=(0.9, 0.95), eps=1e-8, grad_scale=1.0):
        self._adam_state()
        mW, vW, mP, vP = self.adam
        check(self.lib.mebt_adamw_step(self.h, ptr(mW), ptr(vW), ptr(mP), ptr(vP), float(lr), float(betas[0]),
                                       float(betas[1]), float(eps), float(weight_decay), int(step), float(grad_scale),
                                       cur_stream()))

    # ---- layout helpers ------------------------------------------------------------------------------
    def views(self, shapes, grads=False):
        """{state-dict name: view into the flat buffers} for the given {name: shape}."""
        Wn, Pn = flat_layout(self.n_layer, has_sos=self.n_latent > 0)
        out = {}
        for names, flat in ((Wn, self.gW if grads else self.W), (Pn, self.gP if grads else self.P)):
            off = 0
            for n in names:
                numel = int(math.prod(shapes[n]))
                out[n] = flat[off:off + numel].view(*shapes[n])
                off += numel
            assert off == flat.numel(), (off, flat.numel())
        return out

    def layer_w_range(self, hi, lo):
        """[start, end) element range in W / gW covering layers lo..hi."""
        per = 12 * self.n_embd * self.n_embd
        return lo * per, (hi + 1) * per

    def head_w_range(self):
        return self.n_layer * 12 * self.n_embd * self.n_embd, self.n_w

    def layer_p_range(self, hi, lo):
        """[start, end) element range in P / gP of the LN-affine and bias slices of layers lo..hi."""
        per = 13 * self.n_embd
        return lo * per, (hi + 1) * per

    def tail_p_range(self):
        """ln_f + mask/sos/pos/tok embeddings"""
        return self.n_layer * 13 * self.n_embd, self.n_p
