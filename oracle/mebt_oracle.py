"""CPU oracle for the MeBT transformer hot path — a from-scratch *restatement* of the reference
algorithm in plain functional PyTorch (fp32, CPU).

TEST INFRASTRUCTURE ONLY.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this module; the shipped path (`mebt_amd/`, `libmebt_hip.so`) never does and
fails loudly when the HIP library is missing.

Parity pin: the reference ships no tests / golden vectors, so this oracle is pinned
against the *imported reference itself*: `tests/golden/make_golden.py` runs the reference's
`Net2NetTransformer` (with the 5 import stubs of SURVEY.md §8c) on closed-form weights
(`oracle/closed_form.py`) and commits inputs + reduced outputs under `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function below against them.

Conventions
* `P` is a dict {state-dict name -> fp32 tensor} using the reference's parameter names
  (SURVEY.md §A.2), e.g. `transformer.blocks.3.attn.query.weight`.
* All randomness is injected: `t`, permutations and Exp(1)/Normal noise are explicit arguments.
* Dropout sites (gpt.py:113-114,135,140,154,211,238-241) are the identity by default: the oracle
  states the p=0 / eval-mode function, which is what every parity test compares.  To check that the
  HIP kernels put their (counter-based) masks at the reference's sites, the network functions take
  an optional `drop(kind, layer, tensor)` hook that multiplies explicit keep-scale masks in at
  exactly those sites: kind in {"emb_sos","emb_ctx","emb_tgt"} (gpt.py:238-240), "attn" (:135),
  "proj" (:140), "mlp" (:154).

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5          # nn.LayerNorm default, mebt/modules/gpt.py:147-148,216


# --------------------------------------------------------------------------------------------
# configuration helpers
# --------------------------------------------------------------------------------------------
class OracleConfig:
    """Plain holder for the handful of hyper-parameters the path needs
    (configs/stl/mebt_16f.yaml:4-57 -> mebt/transformer.py:73-142)."""

    def __init__(self, n_layer, n_head, n_embd, block_size, sos_emb, mode, vocab_size=16384,
                 shape=(4, 16, 16), schedule="linear", budget=1024, avg_loss=1.0,
                 label_smoothing=0.0):
        mode = list(mode)
        if len(mode) < n_layer:                       # gpt.py:208-209 pads with 'maskgit'
            mode = mode + ["maskgit"] * (n_layer - len(mode))
        assert len(mode) == n_layer                   # gpt.py:213
        assert n_embd % n_head == 0                   # gpt.py:107
        self.n_layer, self.n_head, self.n_embd = n_layer, n_head, n_embd
        self.block_size, self.sos_emb, self.mode = block_size, sos_emb, mode
        self.vocab_size = vocab_size                  # forced to 16384 under vtokens, transformer.py:192
        self.shape, self.schedule, self.budget = tuple(shape), schedule, budget
        self.avg_loss, self.label_smoothing = float(avg_loss), float(label_smoothing)


def param_shapes(cfg):
    """State-dict schema, SURVEY.md §A.2 (transformer.py:126-140, gpt.py:109-116,147-155,216-217)."""
    d, V = cfg.n_embd, cfg.vocab_size
    s = {"mask_emb": (1, 1, d), "pos_emb": (1, cfg.block_size, d), "tok_emb.weight": (V, d)}
    if cfg.sos_emb > 0:
        s["sos_emb"] = (1, cfg.sos_emb, d)
    for i in range(cfg.n_layer):
        p = f"transformer.blocks.{i}."
        for ln in ("ln1", "ln2"):
            s[p + ln + ".weight"] = (d,)
            s[p + ln + ".bias"] = (d,)
        for lin in ("key", "query", "value", "proj"):
            s[p + f"attn.{lin}.weight"] = (d, d)
            s[p + f"attn.{lin}.bias"] = (d,)
        s[p + "mlp.0.weight"] = (4 * d, d)
        s[p + "mlp.0.bias"] = (4 * d,)
        s[p + "mlp.2.weight"] = (d, 4 * d)
        s[p + "mlp.2.bias"] = (d,)
    s["transformer.ln_f.weight"] = (d,)
    s["transformer.ln_f.bias"] = (d,)
    s["transformer.head.weight"] = (V, d)
    return s


def closed_form_params(cfg, requires_grad=False):
    from . import closed_form as cf
    P = {k: torch.from_numpy(v) for k, v in cf.state_dict_numpy(param_shapes(cfg)).items()}
    if requires_grad:
        for v in P.values():
            v.requires_grad_(True)
    return P


# --------------------------------------------------------------------------------------------
# schedules (mebt/mask_sampler.py:34-65) — scalar/tensor in, same type out
# --------------------------------------------------------------------------------------------
def schedule_value(name, t):
    t = torch.as_tensor(t, dtype=torch.float32) if not torch.is_tensor(t) else t
    if name == "cosine":
        return torch.cos(0.5 * np.pi * t)              # :36
    if name == "cosine_plus":
        return 0.5 * (1 + torch.cos(np.pi * t))        # :40
    if name == "linear":
        return 1.0 - t                                 # :45
    if name == "quadratic":
        return (1.0 - t) ** 2.0                        # :49
    if name == "square":
        return 1.0 - t ** 2.0                          # :53
    if name == "cube":
        return 1.0 - t ** 3.0                          # :57
    if name == "sqrt":
        return 1.0 - t ** 0.5                          # :61
    if name == "convex":
        return (1.0 - t) ** 3.0                        # :65
    raise ValueError(name)


def ctemp_factor(name, t_next):
    """module-level linear/constant/cosine of mebt/transformer.py:51-58 (resolved there by eval)."""
    if name == "linear":
        return 1.0 - t_next
    if name == "constant":
        return 1.0
    if name == "cosine":
        return np.cos(t_next * np.pi / 2.0)
    raise ValueError(name)


# --------------------------------------------------------------------------------------------
# mask bookkeeping
# --------------------------------------------------------------------------------------------
def divide_indices(indices, t, cfg, training, window=None):
    """mebt/mask_sampler.py:75-115.  `indices` [B,N] int64 (one permutation per row), `t` scalar.
    `window` = None (T == max_T, the 'longest' prior of transformer.py:46-49) or (T, start_t): the
    two numpy draws of :88,:90 made explicit.  Returns (context [B,NC], target [B,NT], seq_len)."""
    mask_ratio = schedule_value(cfg.schedule, torch.tensor(float(t)))
    if training and window is not None:
        T, start_t = window
        num_pos = int(np.prod(cfg.shape[1:]))
        if T != cfg.shape[0]:                                           # :89
            lo, hi = start_t * num_pos, (start_t + T) * num_pos         # :93-94
            rows = [row[(row >= lo) & (row < hi)] for row in indices]   # :97-98 keeps order
            indices = torch.stack(rows)
    seq_len = int(indices.shape[1])                                     # :101
    n_masked = int(torch.ceil(mask_ratio * seq_len).to(torch.long))     # :102
    n_ctx = seq_len - n_masked                                          # :103
    budget = cfg.budget if training else seq_len                        # :105-108
    n_tgt = min(budget, seq_len - n_ctx)                                # :111
    return indices[:, :n_ctx], indices[:, -n_tgt:], seq_len             # :113-114 (note -0: quirk)


# --------------------------------------------------------------------------------------------
# network
# --------------------------------------------------------------------------------------------
def embed(P, cfg, x_ids, ci, ti):
    """mebt/transformer.py:255-277 / :298-317.  x_ids [B,N] i64; ci [B,NC], ti [B,NT] i64.
    contexts = tok_emb[x[ci]] + pos_emb[ci]; targets = mask_emb + pos_emb[ti]; sos broadcast."""
    B = x_ids.shape[0]
    z_ctx = torch.gather(x_ids, 1, ci)                                   # :255
    pos = P["pos_emb"][0]                                                # [block, d]
    contexts = P["tok_emb.weight"][z_ctx] + pos[ci]                      # :262,:269,:271
    targets = P["mask_emb"].reshape(1, 1, -1) + pos[ti]                  # :263,:270,:272
    if cfg.sos_emb > 0:
        sos = P["sos_emb"].expand(B, -1, -1)                             # :274
    else:
        sos = torch.zeros(B, 0, cfg.n_embd)                              # :276
    return sos, contexts, targets


def layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def _nodrop(kind, layer, t):
    return t


def cross_attention(P, pre, n_head, query, key, drop=_nodrop, layer=0):
    """mebt/modules/gpt.py:119-141 (attn_bias == 0.0 at every call site, transformer.py:281,321)."""
    B, NQ, C = query.shape
    NK = key.shape[1]
    hd = C // n_head
    lin = lambda name, x: F.linear(x, P[pre + f"attn.{name}.weight"], P[pre + f"attn.{name}.bias"])
    k = lin("key", key).view(B, NK, n_head, hd).transpose(1, 2)          # :126
    q = lin("query", query).view(B, NQ, n_head, hd).transpose(1, 2)      # :127
    v = lin("value", key).view(B, NK, n_head, hd).transpose(1, 2)        # :128
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd))              # :131 scale after the product
    att = F.softmax(att, dim=-1)                                         # :134
    att = drop("attn", layer, att)                                       # :135 attn_drop
    y = (att @ v).transpose(1, 2).contiguous().view(B, NQ, C)            # :136-137
    return drop("proj", layer, lin("proj", y))                           # :140 resid_drop(proj(y))


def block(P, i, mode, n_head, sos, ctx, tgt, drop=_nodrop):
    """mebt/modules/gpt.py:159-195.  Quirks kept: the residual is taken on the
    *normalised* query (:180,:184) and ln1 is shared by query and key (:180-181)."""
    pre = f"transformer.blocks.{i}."
    NC = ctx.shape[1]
    if mode == "latent_self":
        query, key = sos, sos                                            # :165-166
    elif mode == "latent_enc":
        query, key = sos, ctx                                            # :168-169
    elif mode == "latent_dec":
        query, key = tgt, sos                                            # :171-172
    elif mode == "lt2l":
        query, key = sos, torch.cat([sos, tgt], 1)                       # :174-175
    elif mode == "maskgit":
        query = torch.cat([ctx, tgt], 1)                                 # :177-178
        key = query
    else:
        raise AssertionError(mode)
    qn = layer_norm(query, P[pre + "ln1.weight"], P[pre + "ln1.bias"])   # :180
    kn = layer_norm(key, P[pre + "ln1.weight"], P[pre + "ln1.bias"])     # :181
    x = qn + cross_attention(P, pre, n_head, qn, kn, drop, i)            # :182,:184
    h = layer_norm(x, P[pre + "ln2.weight"], P[pre + "ln2.bias"])
    h = F.linear(h, P[pre + "mlp.0.weight"], P[pre + "mlp.0.bias"])
    h = F.gelu(h)                                                        # exact erf GELU, :152
    x = x + drop("mlp", i, F.linear(h, P[pre + "mlp.2.weight"], P[pre + "mlp.2.bias"]))  # :185, :154
    if mode in ("latent_enc", "latent_self", "lt2l"):
        sos = x                                                          # :187-188
    elif mode == "latent_dec":
        tgt = x                                                          # :189-190
    else:
        ctx, tgt = x[:, :NC], x[:, NC:]                                  # :191-192
    return sos, ctx, tgt


def gpt_forward(P, cfg, sos, ctx, tgt, return_hidden=False, drop=_nodrop):
    """mebt/modules/gpt.py:234-253 (dropouts are identity unless `drop` is given; head has no bias :217)."""
    hidden = []
    sos, ctx, tgt = drop("emb_sos", 0, sos), drop("emb_ctx", 0, ctx), drop("emb_tgt", 0, tgt)   # :238-240
    for i, mode in enumerate(cfg.mode):
        sos, ctx, tgt = block(P, i, mode, cfg.n_head, sos, ctx, tgt, drop)
        if return_hidden:
            hidden.append((sos, tgt))
    x = layer_norm(tgt, P["transformer.ln_f.weight"], P["transformer.ln_f.bias"])   # :247
    logits = F.linear(x, P["transformer.head.weight"])                                 # :248
    return (logits, hidden) if return_hidden else logits


def reconstruct_mask(P, cfg, x_ids, ci, ti, drop=_nodrop):
    """mebt/transformer.py:288-324: logits [B,NT,V] for caller-supplied index sets."""
    B = x_ids.shape[0]
    x_ids = x_ids.reshape(B, -1)
    sos, ctx, tgt = embed(P, cfg, x_ids, ci, ti)
    return gpt_forward(P, cfg, sos, ctx, tgt, drop=drop)


def forward(P, cfg, x, indices, t, training=True, window=None, drop=_nodrop):
    """mebt/transformer.py:216-286 with the RNG draw `t` (:228) explicit.
    Returns (logits, z_targets, NT_weight, seq_len)."""
    B = x.shape[0]
    x_ids = x.reshape(B, -1)                                              # encode_to_z, :685-686
    ci, ti, seq_len = divide_indices(indices, t, cfg, training, window)   # :251
    z_tgt = torch.gather(x_ids, 1, ti)                                    # :256
    NT_weight = float(seq_len - ci.shape[1])                              # :258-259 (before budget cut)
    logits = reconstruct_mask(P, cfg, x_ids, ci, ti, drop)
    return logits, z_tgt, NT_weight, seq_len


# --------------------------------------------------------------------------------------------
# loss / metrics / optimiser
# --------------------------------------------------------------------------------------------
def loss_and_acc(logits, z_tgt, NT_weight, seq_len, cfg):
    """mebt/transformer.py:717-732 + mebt/utils.py:80-94.  Returns (acc1, acc5, loss)."""
    B, _, V = logits.shape
    ratio = NT_weight / float(seq_len)                                    # :723
    ce = F.cross_entropy(logits.reshape(-1, V), z_tgt.reshape(-1), reduction="sum",
                         label_smoothing=cfg.label_smoothing)             # :726
    loss = ce / (B * seq_len * ratio ** cfg.avg_loss)                     # :729-730
    flat, tg = logits.reshape(-1, V), z_tgt.reshape(-1)
    _, pred = flat.topk(5, 1, True, True)                                 # utils.py:86
    hit = pred.eq(tg.reshape(-1, 1))
    n = tg.numel()
    acc1 = hit[:, :1].float().sum() * (100.0 / n)                         # utils.py:92-93
    acc5 = hit[:, :5].float().sum() * (100.0 / n)
    return acc1, acc5, loss


def decay_split(P):
    """mebt/transformer.py:749-798: (decay, emb, no_decay, pos) name lists = the 4 AdamW groups.
    decay = weights of nn.Linear inside `transformer.*`; embeddings/LN/biases are not decayed."""
    decay, no_decay = [], []
    for name in P:
        if not name.startswith("transformer."):
            continue
        short = name[len("transformer."):]
        if short.endswith("bias"):
            no_decay.append(short)                                        # :766-768
        elif ".ln" in short or short.startswith("ln_f"):
            no_decay.append(short)                                        # :772-774 (LayerNorm)
        else:
            decay.append(short)                                           # :769-771 (Linear)
    emb = [n for n in P if "_emb" in n and n != "pos_emb"]                # :777-778
    pos = [n for n in P if "pos_emb" in n]                                # :779
    return sorted(decay), emb, sorted(no_decay), pos                      # :791-794 (sorted lists)


def adamw_update(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.95, eps=1e-8):
    """torch.optim.AdamW single-tensor update (decoupled decay), as configured at
    mebt/transformer.py:797 (betas (0.9,0.95), default eps 1e-8).  In-place on p, m, v."""
    p.mul_(1.0 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def lr_at(global_step, base_lr, warmup_steps, cosine_lr, max_steps):
    """mebt/transformer.py:665-678 (the LR that `optimizer_step` installs before `step()`)."""
    if global_step < warmup_steps:
        return base_lr * min(1.0, float(global_step + 1) / warmup_steps)
    if cosine_lr:
        rad = float(global_step - warmup_steps) / float(max_steps - warmup_steps)
        return base_lr * 0.5 * (1 + np.cos(rad * np.pi))
    return base_lr


class TrainState:
    """Parameters + AdamW moments for `train_step` (counterpart of the Lightning fit loop,
    SURVEY.md §3.1: shared_step -> backward -> optimizer_step)."""

    def __init__(self, P, lr, weight_decay=0.01, warmup_steps=0, cosine_lr=False, max_steps=0):
        self.P = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
        self.m = {k: torch.zeros_like(v) for k, v in P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in P.items()}
        self.lr, self.wd = lr, weight_decay
        self.warmup_steps, self.cosine_lr, self.max_steps = warmup_steps, cosine_lr, max_steps
        self.global_step = 0
        decay, _, _, _ = decay_split(P)
        self.decay_names = {"transformer." + n for n in decay}


def train_step(state, cfg, x, indices, t, window=None, grad_hook=None, drop=_nodrop):
    """One optimiser step: forward (transformer.py:216-286), loss (:717-732), backward, AdamW
    (:665-681,:790-797).  `grad_hook(grads)` lets the DP tests average gradients across ranks
    before the update (train_transformer.py:39-41 DDP).  Returns dict(loss, acc1, acc5, grads)."""
    P = state.P
    for p in P.values():
        p.grad = None
    logits, z_tgt, NT_weight, seq_len = forward(P, cfg, x, indices, t, training=True, window=window, drop=drop)
    acc1, acc5, loss = loss_and_acc(logits, z_tgt, NT_weight, seq_len, cfg)
    loss.backward()
    # a parameter the loss does not depend on (e.g. an `lt2l` block placed last) has grad None and
    # torch.optim.AdamW skips it entirely — no decay, no moment update; keep that behaviour.
    has_grad = {k for k, p in P.items() if p.grad is not None}
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in P.items()}
    if grad_hook is not None:
        grad_hook(grads)
    lr = lr_at(state.global_step, state.lr, state.warmup_steps, state.cosine_lr, state.max_steps)
    state.global_step += 1
    with torch.no_grad():
        for k, p in P.items():
            if k not in has_grad:
                continue
            wd = state.wd if k in state.decay_names else 0.0
            adamw_update(p, grads[k], state.m[k], state.v[k], state.global_step, lr, wd)
    return {"loss": float(loss.detach()), "acc1": float(acc1), "acc5": float(acc5), "grads": grads,
            "logits": logits.detach(), "n_targets": int(z_tgt.numel())}


# --------------------------------------------------------------------------------------------
# sampler
# --------------------------------------------------------------------------------------------
def top_k_logits(logits, k):
    """mebt/transformer.py:891-895: keep everything >= the k-th largest value (ties kept)."""
    v, _ = torch.topk(logits, k)
    out = logits.clone()
    out[out < v[..., [-1]]] = -float("inf")
    return out


def top_p_probs(probs, p):
    """mebt/transformer.py:898-910: drop tokens *after* the first whose cumulative prob >= p,
    then renormalise."""
    sp, si = torch.sort(probs, dim=-1, descending=True)
    remove = torch.cumsum(sp, dim=-1) >= p
    remove[..., 1:] = remove[..., :-1].clone()
    remove[..., 0] = False
    remove = remove.scatter(-1, si, remove)
    probs = probs.masked_fill(remove, 0.0)
    return probs / probs.sum(-1, keepdim=True)


def sample_from_logits(logits, temperature, top_k, top_p, noise):
    """mebt/transformer.py:843-889 + gumbel_sort :826-841 with the Exp(1) draw (:837) injected as
    `noise` (same shape as logits).  Returns (ids [..], probs [..,V]).  The reference sorts the
    whole vocabulary and takes column 0 (:839,:877); that is arg-max of p_norm/q with zero-prob
    entries forced to 0 (:838)."""
    logits = logits.to(torch.float32) / (temperature + 1e-8)             # :859-860
    if top_k is not None:
        logits = top_k_logits(logits, top_k)                              # :863-864
    logits = torch.where(torch.isnan(logits), torch.full_like(logits, -float("inf")), logits)  # :866-868
    probs = F.softmax(logits, dim=-1)                                     # :871
    if top_p is not None:
        probs = top_p_probs(probs, top_p)                                 # :873-874
    pn = probs / probs.sum(-1, keepdim=True)                              # :834
    key = (pn / noise) * (pn > 0).float()                                 # :835-838
    ids = key.sort(dim=-1, descending=True)[1][..., 0]                    # :839,:877
    return ids, probs


def inverse_cdf_order(V=16384):
    """element order of the product's inverse-CDF draw (mebt_amd/csrc/sampler.hip: sample_fast_kernel, `draw = 1`): thread t of 512
    owns elements 8 t + k + 4096 g (g = 0..3 major, k = 0..7), threads in order."""
    assert V == 16384
    return (8 * torch.arange(512).view(512, 1, 1) + torch.arange(8).view(1, 1, 8) + 4096 * torch.arange(4).view(1, 4, 1)).reshape(-1)


def sample_inverse_cdf(logits, temperature, top_k, seed):
    """CPU twin of the product's PRODUCTION draw (no injected noise; include/mebt_hip.h: mebt_op_sample_lp, draw = 1).  The
    reference draws arg-max p / q, q ~ Exp(1) (mebt/transformer.py:826-841, :877) - one sample of the categorical distribution p
    left by temperature / top-k / softmax (:859-871).  The product takes the same distribution's sample by inverse CDF from ONE
    uniform per row: the first element, in `inverse_cdf_order`, whose running sum of p reaches u * sum(p), u = the counter-based
    uniform of (seed, row) (oracle/closed_form.py:uniform_counter, exact twin of the kernel's hash); filtered elements (p = 0) are
    never chosen; if rounding leaves the target above the last running sum the last positive element is taken.
    Probabilities: this oracle's own (`sample_from_logits`: the reference's arithmetic), running sum in float64.
    Returns (ids [R], probs [R, V], margin [R]) for logits [R, V]; margin = distance of the target from the nearer running-sum
    boundary of the chosen element, relative to sum(p): a product id that differs is a proven tie only where the margin is within
    fp32 rounding of the running sum (the product adds 32 elements per thread, then scans 512 thread sums, in fp32)."""
    from oracle import closed_form as cf
    R, V = logits.shape
    _, probs = sample_from_logits(logits, temperature, top_k, None, torch.ones(R, V))
    order = inverse_cdf_order(V)
    po = probs.double()[:, order]
    cum = po.cumsum(1)
    tot = cum[:, -1]
    u = torch.from_numpy(cf.uniform_counter(int(seed), R).astype(np.float64))
    target = u * tot
    ids = torch.empty(R, dtype=torch.long)
    margin = torch.empty(R, dtype=torch.float64)
    for r in range(R):
        pos = (po[r] > 0).nonzero().flatten()
        hit = ((cum[r] >= target[r]) & (po[r] > 0)).nonzero().flatten()
        e = int(hit[0]) if hit.numel() else int(pos[-1])
        ids[r] = order[e]
        lo = float(cum[r, e] - po[r, e])
        margin[r] = min(abs(float(target[r]) - lo), abs(float(cum[r, e]) - float(target[r]))) / float(tot[r])
    return ids, probs, margin


def gumbel_top_k(score, ctemp, noise):
    """mebt/mask_sampler.py:178-187 with the Exp(1) draw injected.  Returns descending order."""
    prob = score / score.sum(-1, keepdim=True)
    prob = prob / (noise ** ctemp)
    return prob.sort(dim=-1, descending=True)[1]


def generate_next_mask(ci, ti, score, n_masked_row0, strategy, ctemp, noise, randn=None):
    """mebt/mask_sampler.py:189-237 for strategies maskgit/random/mlm/bootstrap.
    `n_masked_row0` = n_masked_toks[0] (:216); `noise` Exp(1) [B,NT] — a tensor, or a callable
    drawn only when the reference would draw it (after the early return :222-225); `randn`
    replaces torch.randn_like(score) for 'random'/'bootstrap' (:206-208)."""
    B, NC = ci.shape
    NT = ti.shape[1]
    if strategy in ("random", "bootstrap"):
        score, ctemp = randn, 0.0                                         # :206-208
    seq_len = NC + NT                                                     # :210
    n_masked = int(n_masked_row0)
    if strategy == "bootstrap":
        n_masked = NT - 1                                                 # :218-219
    n_ctx = seq_len - n_masked                                            # :220
    if n_ctx <= NC:
        return ci, ti                                                     # :222-225
    n_new = n_ctx - NC                                                    # :227
    if callable(noise):
        noise = noise()
    order = gumbel_top_k(score, ctemp, noise)                             # :229
    new_ctx = torch.cat([ci, torch.gather(ti, -1, order[:, :n_new])], 1)  # :228,:232-233
    new_tgt = torch.gather(ti, -1, order[:, n_new:])                      # :231,:234 (score order)
    return new_ctx, new_tgt


def scatter_ids(partial, ti, ids):
    """mebt/transformer.py:413-439 (and :571-585, :615-629): write `ids` at positions `ti` of each
    row.  The reference builds two sparse COO tensors + torch.where; index sets are duplicate-free
     so it is a plain row-wise scatter."""
    out = partial.clone()
    out.scatter_(1, ti, ids)
    return out


def sample(P, cfg, x, n_steps, temperature, top_k, top_p, context_temperature, noise_fn,
           strategy="maskgit", ctemp_schedule="linear", schedule=None, ci=None, ti=None,
           edit=False, logits_fn=None, trace=None, return_probs=False):
    """mebt/transformer.py:353-447.  `noise_fn(tag, shape)` supplies every random draw in call
    order: tags 'sample' (:407 -> :837), 'mask' (:444 -> mask_sampler.py:182), 'randn'
    (mask_sampler.py:207).  `logits_fn(x_ids, ci, ti)` overrides the network (used to drive the
    oracle's bookkeeping from HIP logits).  Returns (x [B,N], ci, ti), plus the `debug=True`
    probability map [B,N,V] (-1 where never sampled, :395,:426-436) with `return_probs`."""
    B = x.shape[0]
    N = int(np.prod(x.shape[1:]))
    edit_N = ti.shape[1] if edit else N                                   # :373-376
    x = x.reshape(B, N)
    sched = schedule or cfg.schedule
    if ci is None:
        ci = torch.empty(B, 0, dtype=torch.long)                          # :385
        ti = torch.arange(N).repeat(B, 1)                                 # :386
    else:
        ci, ti = ci.clone(), ti.clone()
    if logits_fn is None:
        logits_fn = lambda xi, c, t_: reconstruct_mask(P, cfg, xi, c, t_)
    partial = x
    partial_probs = -torch.ones(B, N, cfg.vocab_size) if return_probs else None   # :395
    for t_next in np.linspace(0, 1, n_steps + 1)[1:]:                     # :391,:397
        tt = torch.full((B,), fill_value=t_next)                          # :398 (float32 on purpose)
        n_masked = torch.ceil(schedule_value(sched, tt) * edit_N)         # :399
        if int((n_masked > ti.shape[-1]).sum()) == B:                     # :401-402
            continue
        logits = logits_fn(partial, ci, ti)                               # :403
        ids, probs = sample_from_logits(logits, temperature, top_k, top_p,
                                        noise_fn("sample", logits.shape))  # :407
        scores = probs.gather(-1, ids.unsqueeze(-1)).squeeze(-1)          # :409
        partial = scatter_ids(partial, ti, ids)                           # :413-439
        if return_probs:                                                  # :426-436
            partial_probs.scatter_(1, ti.unsqueeze(-1).expand(-1, -1, probs.shape[-1]), probs)
        ctemp = context_temperature * ctemp_factor(ctemp_schedule, t_next)  # :440
        if trace is not None:
            trace.append({"NC": ci.shape[1], "NT": ti.shape[1], "ids": ids.clone(), "ci": ci.clone(), "ti": ti.clone(),
                          "scores": scores.clone(), "ctemp": float(ctemp), "n_masked": int(n_masked[0])})
        rn = noise_fn("randn", scores.shape) if strategy in ("random", "bootstrap") else None
        ci, ti = generate_next_mask(ci, ti, scores, n_masked[0].long(), strategy, ctemp,
                                    lambda: noise_fn("mask", scores.shape), rn)   # :444
    if return_probs:
        return partial.view(B, -1), ci, ti, partial_probs
    return partial.view(B, -1), ci, ti


def bidirect_sample(P, cfg, batch_size, total_length, step_size, context_size, temperature, top_k, top_p,
                    vid_n_steps, vid_c_temp, noise_fn, ctemp_schedule="linear", strategy="maskgit", bootstrap=0, logits_fn=None):
    """sample_vqgan_transformer_videos.py:22-94 without the VQGAN decode: returns (code_map
    [B,T',H,W], score [B]).  The score gathers over the first window only (the reference's gather
    :89-92 only type-checks when no sliding-window continuation happened)."""
    T, H, W = cfg.shape
    step, ctx = int(step_size * 0.25), int(context_size * 0.25)           # :29-31
    shape = (batch_size, step, H, W)
    x = torch.zeros(shape, dtype=torch.long)
    ci = ti = boot = None
    if bootstrap > 0:                                                     # :41-42
        x, ci, ti, boot = sample(P, cfg, x, bootstrap, 1., None, None, vid_c_temp, noise_fn, strategy="bootstrap",
                                 ctemp_schedule=ctemp_schedule, ci=ci, ti=ti, return_probs=True, logits_fn=logits_fn)
    x, ci, _, final = sample(P, cfg, x.reshape(shape), vid_n_steps, temperature, top_k, top_p, vid_c_temp, noise_fn,
                             strategy=strategy, ctemp_schedule=ctemp_schedule, ci=ci, ti=ti, return_probs=True, logits_fn=logits_fn)   # :43-46
    vq = x.reshape(shape)
    code_map, curr_t = [vq], step
    while curr_t < total_length * 0.25:                                   # :55-71
        new_x = torch.zeros(shape, dtype=torch.long)
        new_x[:, :ctx] = vq[:, -ctx:]
        ci = torch.arange(H * W * ctx).repeat(batch_size, 1)
        ti = torch.arange((step - ctx) * H * W).repeat(batch_size, 1) + H * W * ctx
        x = sample(P, cfg, new_x, vid_n_steps, temperature, top_k, top_p, vid_c_temp, noise_fn, strategy=strategy,
                   ctemp_schedule=ctemp_schedule, ci=ci, ti=ti, logits_fn=logits_fn)[0]
        vq = x.reshape(shape)
        code_map.append(vq[:, ctx:])
        curr_t += step - ctx
    code_map = torch.cat(code_map, 1)
    prob_map = final if boot is None else torch.where(final < 0., boot, final)        # :86-88
    first = code_map.reshape(batch_size, -1)[:, :prob_map.shape[1]]
    score = torch.gather(prob_map, -1, first.unsqueeze(-1)).squeeze(-1).log().sum(-1)  # :89-92
    return code_map, score


def extrapolate(P, cfg, vq_input, total_length, step_size, context_size, temperature, top_k, top_p,
                vid_n_steps, vid_c_temp, noise_fn, logits_fn=None):
    """sample_vqgan_transformer_videos.py:96-157 without the VQGAN decode: returns the code map."""
    B, T, H, W = vq_input.shape
    step, ctx = int(step_size * 0.25), int(context_size * 0.25)
    assert T == step                                                      # :106
    jump = step - ctx
    n_jumps = int(np.ceil((int(total_length * 0.25) - step) / jump))      # :108-112
    idx = torch.arange(H * W * step).repeat(B, 1).view(B, step, H, W)
    ci, ti = idx[:, :ctx].reshape(B, -1), idx[:, ctx:].reshape(B, -1)     # :128-131
    code_map, x = [vq_input.clone()], vq_input
    for _ in range(n_jumps):                                              # :135-145
        nxt = torch.zeros_like(x)
        nxt[:, :ctx] = code_map[-1][:, -ctx:]
        x = sample(P, cfg, nxt.view(B, -1), vid_n_steps, temperature, top_k, top_p, vid_c_temp, noise_fn,
                   ci=ci, ti=ti, edit=True, logits_fn=logits_fn)[0].view(B, step, H, W)
        code_map.append(x[:, ctx:].clone())
    return torch.cat(code_map, 1)


def gibbs_revise_masks(ci, ti, n_steps, perms):
    """mebt/mask_sampler.py:317-336; `perms` [B,N] replaces the per-sample torch.randperm (:331)."""
    N = ti.shape[1]
    assert N % n_steps == 0                                               # :328
    w = N // n_steps
    ti = torch.gather(ti, 1, perms)                                       # :332
    ctxs = [torch.cat([ci, ti[:, (i + 1) * w:], ti[:, :i * w]], 1) for i in range(n_steps)]   # :334
    tgts = [ti[:, i * w:(i + 1) * w] for i in range(n_steps)]             # :335
    return ctxs, tgts


def gibbs_draft_masks(ci, ti, n_steps, perms):
    """mebt/mask_sampler.py:338-356; `perms` replaces torch.randperm (:351)."""
    N = ti.shape[1]
    assert N % n_steps == 0                                               # :348
    w = N // n_steps
    ti = torch.gather(ti, 1, perms)                                       # :352
    ctxs = [torch.cat([ci, ti[:, :i * w]], 1) for i in range(n_steps)]    # :354
    tgts = [ti[:, i * w:] for i in range(n_steps)]                        # :355
    return ctxs, tgts


def _gibbs_pass(P, cfg, x, masks, temperature, top_k, top_p, noise_fn, logits_fn):
    """Shared body of draft (:565-586) and revise (:609-630)."""
    partial = x
    for c, t_ in zip(*masks):
        logits = logits_fn(partial, c, t_)
        ids, _ = sample_from_logits(logits, temperature, top_k, top_p, noise_fn("sample", logits.shape))
        partial = scatter_ids(partial, t_, ids)
    return partial


def draft_and_revise(P, cfg, x, n_draft, draft_t, draft_k, draft_p, n_revise, revise_t, revise_k,
                     revise_p, M, skip_draft, perm_fn, noise_fn, logits_fn=None):
    """mebt/transformer.py:632-663 (+ draft :544-586, revise :588-630).  `perm_fn(tag, B, N)`
    supplies the [B,N] permutations of mask_sampler.py:331,351 in call order."""
    B = x.shape[0]
    N = int(np.prod(x.shape[1:]))
    x = x.reshape(B, N)
    if logits_fn is None:
        logits_fn = lambda xi, c, t_: reconstruct_mask(P, cfg, xi, c, t_)
    ci0 = torch.empty(B, 0, dtype=torch.long)                             # :559 / :603
    ti0 = torch.arange(N).repeat(B, 1)                                    # :560 / :604
    if not skip_draft:                                                    # :655-656
        masks = gibbs_draft_masks(ci0, ti0, n_draft, perm_fn("draft", B, N))
        x = _gibbs_pass(P, cfg, x, masks, draft_t, draft_k, draft_p, noise_fn, logits_fn)
    for _ in range(M):                                                    # :661-662
        masks = gibbs_revise_masks(ci0, ti0, n_revise, perm_fn("revise", B, N))
        x = _gibbs_pass(P, cfg, x, masks, revise_t, revise_k, revise_p, noise_fn, logits_fn)
    return x.view(B, -1)                                                  # :663


# --------------------------------------------------------------------------------------------
# FLOP model (validated there against torch.utils.flop_counter)
# --------------------------------------------------------------------------------------------
def forward_flops_per_sample(cfg, NC, NT):
    d, NS, V = cfg.n_embd, cfg.sos_emb, cfg.vocab_size
    total = 0
    for mode in cfg.mode:
        NQ, NK = {"latent_enc": (NS, NC), "latent_self": (NS, NS), "latent_dec": (NT, NS),
                  "lt2l": (NS, NS + NT), "maskgit": (NC + NT, NC + NT)}[mode]
        total += (2 * NQ + 2 * NK) * d * d + 2 * NQ * NK * d + 8 * NQ * d * d
    total += NT * d * V
    return 2 * total
