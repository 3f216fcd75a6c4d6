"""Parity of the BENCHMARKED path at the BENCHMARKED size.

* C2 (Sky-Timelapse 16f, 24L/1024d, batch 6, t = 0.5, bf16, autotuned tiles, pair / grouped launches): one whole
  `TrainLoop.step` — loss, every parameter gradient, AdamW moments and post-step parameters — against one
  `oracle.train_step` on the same weights and batch, without dropout and with the kernels' own dropout masks
  (p = 0.1, as the config trains) injected into the oracle; the optimizer-in-backward step (what bench.py times on
  one GPU) against the separate-optimizer step.
* C4 (UCF-101 128f geometry, block 8192): one revise forward at (NC, NT) = (7936, 256) against
  `oracle.reconstruct_mask` (fp32 <= 1e-3, bf16 at a measured bound) and one whole `draft_and_revise` at block 8192
  with injected noise, token ids bit-exact in fp32 (reference mebt/transformer.py:216-286,632-663,717-732).

GPU only.  The oracle legs are the slow part (~4 s per C2 train step, ~2 s per C4 forward on the box's host cores).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mebt_amd import _lib, presets
from mebt_amd.trainer import TrainLoop
from oracle import mebt_oracle as orc

DEV = "cuda"
# The default run has to fit the driver's time limit with room to spare (617 s of 1200).  With the oracle
# side's CPU threads bounded (tests/conftest.py: 128 default threads ran 5 x slower than 16 on the GPU boxes) the whole suite with
# every sibling case of the oracle-checked full-size families takes ~6 min, so they run by default again; MEBT_LONG_TESTS=0 keeps
# ONE case per family (the benchmarked precision / the hardest shape: ~4 min), MEBT_LONG_TESTS=1 adds the ≈20-minute UCF schedule.
LONG = os.environ.get("MEBT_LONG_TESTS", "") != "0"
long_only = pytest.mark.skipif(not LONG, reason="MEBT_LONG_TESTS=0: only one case per full-size family")
SITE = {"attn": 0, "proj": 1, "mlp": 2, "emb_sos": 0xFFFF0, "emb_ctx": 0xFFFF1, "emb_tgt": 0xFFFF2}

# bf16 (MFMA bf16, fp32 accumulate) against the fp32 oracle at C2, B = 6.  Bounds are <= 2x what this test measured
# on MI355X (printed by the test with -s): logits 2.85e-2 abs at max |logits| = 3.41 (0.84 %; these weights carry
# N(0, 0.02) biases and perturbed LN gains, so the logits are 3-5x larger than with the default init), gradients
# 4.3e-2 of each tensor's max |g|.
BF16_LOGITS_REL = 1.7e-2      # x max |logits| of the oracle
C2_BF16_GRAD = 8e-2
C2_BF16_LOSS_REL = 1e-3
# A max-norm bound of 8 % passes a tensor that is 5 % wrong everywhere: per tensor also the direction
# (cosine) and the relative L2 error against the oracle.  Bounds <= 2x measured on MI355X (printed with -s).
C2_BF16_COS = 0.999
C2_BF16_RELL2 = 4e-2
C2_F32_GRAD = 1e-3            # fp32 engine: every gradient within 1e-3 of its tensor's max (north_star tolerance)
# block 8192 (C4 train step, B = 2: reductions over 8x the tokens of C2 through bf16 activations): measured on MI355X worst
# 1 - cosine 1.09e-3, worst relative L2 4.7e-2, worst max error 5.7e-2 of the tensor's max (profiles/r04_parity_measured.txt);
# gates <= 2x measured.  The fp32 engine on the same step: 1.1e-5 / 3e-11 / 7.8e-6.
C4_BF16_COS = 0.998
C4_BF16_RELL2 = 8e-2


def oracle_cfg_of(cfg):
    p, m = cfg.model.params, cfg.model.mask.params
    return orc.OracleConfig(p.n_layer, p.n_head, p.n_embd, p.block_size, p.sos_emb, p.mode, shape=m.shape,
                            schedule=m.schedule, budget=m.budget, avg_loss=1.0)


def kernel_mask(seed, site, p, shape):
    n = int(np.prod(shape))
    pad = (-n) % 4
    out = torch.ones(n + pad, device=DEV)
    _lib.check(_lib.load().mebt_debug_dropout_mask(seed, site, p, n + pad, _lib.ptr(out), _lib.cur_stream()))
    torch.cuda.synchronize()
    return out[:n].cpu().view(*shape)


def perturbed_state(seed, cfg):
    """random-init weights with non-trivial biases / LN affine so that no term is silently zero"""
    torch.manual_seed(seed)
    model = presets.build_model(cfg, compute_dtype="bf16")
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.02)
            elif ".ln" in n and n.endswith("weight"):
                p.add_(torch.randn_like(p) * 0.05)
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def batch(B, shape, seed):
    g = torch.Generator().manual_seed(seed)
    N = int(np.prod(shape))
    x = torch.randint(0, 16384, (B, *shape), generator=g)
    idx = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
    return x, idx


@pytest.mark.parametrize("dropout", [pytest.param(0.0, marks=long_only), 0.1])
def test_c2_bf16_train_step_vs_oracle(dropout):
    cfg = presets.sky_16f(dropout=dropout)
    lr = cfg.exp.exact_lr
    ocfg = oracle_cfg_of(cfg)
    sd = perturbed_state(21, cfg)
    B, t = 6, 0.5
    x, idx = batch(B, [4, 16, 16], 77)

    def fresh(fused):
        m = presets.build_model(cfg, compute_dtype="bf16")
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        return m, TrainLoop(m, fused_optimizer=fused)

    # ---- HIP, separate optimizer: gradients are observable
    model, loop = fresh(False)
    assert not loop.fused_optimizer
    seed = (model.global_step << 20) ^ (model._seed_ctr + 1)            # what TrainLoop.step is about to use
    stats = loop.step(x.to(DEV), idx.to(DEV), t=t).cpu()
    nm = loop.native
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    g_hip = {k: v.detach().cpu().clone() for k, v in nm.views(shapes, grads=True).items()}
    p_hip = {k: v.detach().cpu().clone() for k, v in nm.views(shapes).items()}
    mW, vW, mP, vP = [a.clone() for a in nm.adam]

    # ---- oracle: same weights / batch / t (and the kernels' own masks when dropout is on)
    pk = {"attn": dropout, "proj": dropout, "mlp": dropout, "emb_sos": dropout, "emb_ctx": dropout, "emb_tgt": dropout}

    def drop(kind, layer, tensor):
        site = SITE[kind] if kind.startswith("emb") else 16 * layer + SITE[kind]
        return tensor * kernel_mask(seed, site, pk[kind], tuple(tensor.shape))

    st = orc.TrainState(sd, lr=lr)
    r = orc.train_step(st, ocfg, x, idx, t, drop=drop if dropout > 0 else orc._nodrop)
    assert r["n_targets"] == int(stats[3]) == B * 512
    loss_hip = float(stats[4])
    assert abs(loss_hip - r["loss"]) < C2_BF16_LOSS_REL * abs(r["loss"]), (loss_hip, r["loss"])

    worst = ("", 0.0)
    bad = []
    worst_cos, worst_l2 = ("", 1.0), ("", 0.0)
    for k, ref in r["grads"].items():
        # attn.key.bias has a mathematically zero gradient (softmax is shift invariant; the oracle holds ~1e-13 of fp32
        # rounding there): its bf16 rounding noise is judged on the scale of the same block's query-bias gradient
        scale_of = k.replace("attn.key.bias", "attn.query.bias")
        denom = r["grads"][scale_of].abs().max().item() + 1e-12
        err = (g_hip[k] - ref).abs().max().item() / denom
        if err > worst[1]:
            worst = (k, err)
        if not err < C2_BF16_GRAD:
            bad.append((k, round(err, 4), denom))
        if "attn.key.bias" in k:
            continue                                  # a zero gradient has no direction
        a, b = g_hip[k].double().reshape(-1), ref.double().reshape(-1)
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        rl2 = float((a - b).norm() / (b.norm() + 1e-300))
        if cos < worst_cos[1]:
            worst_cos = (k, cos)
        if rl2 > worst_l2[1]:
            worst_l2 = (k, rl2)
        if not (cos >= C2_BF16_COS and rl2 <= C2_BF16_RELL2):
            bad.append((k, "cos", round(cos, 6), "rel-L2", round(rl2, 5)))
    print(f"[c2 bf16 dropout={dropout}] loss hip {loss_hip:.6f} oracle {r['loss']:.6f}; worst gradient {worst[0]} rel-to-max {worst[1]:.3e}; "
          f"worst cosine {worst_cos[0]} {worst_cos[1]:.6f}; worst rel-L2 {worst_l2[0]} {worst_l2[1]:.3e}")
    from tests.helpers import record_measured
    record_measured(f"c2 train bf16 dropout={dropout}: worst gradient rel-to-max", worst[1], C2_BF16_GRAD, worst[0])
    record_measured(f"c2 train bf16 dropout={dropout}: worst 1-cosine", 1 - worst_cos[1], 1 - C2_BF16_COS, worst_cos[0])
    record_measured(f"c2 train bf16 dropout={dropout}: worst rel-L2", worst_l2[1], C2_BF16_RELL2, worst_l2[0])
    record_measured(f"c2 train bf16 dropout={dropout}: loss rel", abs(loss_hip - r["loss"]) / abs(r["loss"]), C2_BF16_LOSS_REL)
    assert not bad, (len(bad), bad[:30])

    # ---- post-step parameters.  At step 1 AdamW moves every parameter by ~lr * sign(g): where the oracle's |g| is
    # well above bf16 noise the parameter must match tightly; elsewhere within 2 lr (a sign flip of a ~0 gradient)
    for k, ref in st.P.items():
        d = (p_hip[k] - ref.detach()).abs()
        assert d.max().item() <= 2.2 * lr + 1e-7, (k, d.max().item())
        g = r["grads"][k]
        sure = g.abs() > 0.25 * r["grads"][k.replace("attn.key.bias", "attn.query.bias")].abs().max()
        if sure.any():
            assert d[sure].max().item() < 0.12 * lr, (k, d[sure].max().item())
    # moments are linear / quadratic in the gradient: m = 0.1 g, v = 0.05 g^2
    Wn, Pn = _flat_names(nm)
    for names, mflat, vflat in ((Wn, mW, vW), (Pn, mP, vP)):
        off = 0
        for n in names:
            numel = int(np.prod(shapes[n]))
            m_ = mflat[off:off + numel].cpu().view(shapes[n])
            v_ = vflat[off:off + numel].cpu().view(shapes[n])
            off += numel
            g = r["grads"][n]
            gm = r["grads"][n.replace("attn.key.bias", "attn.query.bias")].abs().max().item() + 1e-12
            assert (m_ - 0.1 * g).abs().max().item() < 0.1 * C2_BF16_GRAD * gm, n
            assert (v_ - 0.05 * g * g).abs().max().item() < 0.05 * 2.5 * C2_BF16_GRAD * gm * gm, n

    # ---- the optimizer-in-backward step (bench.py's one-GPU path) == the separate-optimizer step
    del model, loop
    torch.cuda.empty_cache()
    model2, loop2 = fresh(True)
    assert loop2.fused_optimizer
    stats2 = loop2.step(x.to(DEV), idx.to(DEV), t=t).cpu()
    assert abs(float(stats2[4]) - loss_hip) < 1e-6 * abs(loss_hip)
    p_fused = loop2.native.views(shapes)
    worst_p = 0.0
    for k in shapes:
        worst_p = max(worst_p, (p_fused[k].cpu() - p_hip[k]).abs().max().item())
    # the two paths see the same gradient up to fp32 summation order; a ~0 gradient may flip sign (2 lr)
    assert worst_p <= 2.2 * lr, worst_p
    for a, b in zip(loop2.native.adam, (mW, vW, mP, vP)):
        ref = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-3 * ref + 1e-12
    print(f"[c2 bf16 dropout={dropout}] fused vs separate optimizer: max |dp| {worst_p:.3e} (lr {lr:.2e})")


def test_c2_f32_train_step_vs_oracle():
    """The fp32 parity engine at the benchmarked size (its backward was oracle-checked at micro / C1 only,
    and then served as the reference of the ragged-shape test below): one whole train step at C2, batch 6, against
    `oracle.train_step` — loss, EVERY gradient within 1e-3 of its tensor's max, cosine, post-step parameters."""
    cfg = presets.sky_16f(dropout=0.0)
    lr = cfg.exp.exact_lr
    ocfg = oracle_cfg_of(cfg)
    sd = perturbed_state(23, cfg)
    B, t = 6, 0.5
    x, idx = batch(B, [4, 16, 16], 79)
    m = presets.build_model(cfg, compute_dtype="f32")
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    loop = TrainLoop(m, fused_optimizer=False)
    stats = loop.step(x.to(DEV), idx.to(DEV), t=t).cpu()
    nm = loop.native
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    g_hip = {k: v.detach().cpu().clone() for k, v in nm.views(shapes, grads=True).items()}
    p_hip = {k: v.detach().cpu().clone() for k, v in nm.views(shapes).items()}
    st = orc.TrainState(sd, lr=lr)
    r = orc.train_step(st, ocfg, x, idx, t)
    assert abs(float(stats[4]) - r["loss"]) < 2e-5 * abs(r["loss"]), (float(stats[4]), r["loss"])
    worst, worst_cos, bad = ("", 0.0), ("", 1.0), []
    for k, ref in r["grads"].items():
        denom = r["grads"][k.replace("attn.key.bias", "attn.query.bias")].abs().max().item() + 1e-12
        err = (g_hip[k] - ref).abs().max().item() / denom
        worst = max(worst, (k, err), key=lambda v: v[1])
        if not err < C2_F32_GRAD:
            bad.append((k, err))
        if "attn.key.bias" not in k:
            a, b = g_hip[k].double().reshape(-1), ref.double().reshape(-1)
            cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
            worst_cos = min(worst_cos, (k, cos), key=lambda v: v[1])
            if not cos > 1 - 1e-6:
                bad.append((k, "cos", cos))
    print(f"[c2 f32] loss hip {float(stats[4]):.6f} oracle {r['loss']:.6f}; worst gradient {worst[0]} rel-to-max {worst[1]:.3e}; worst cosine {worst_cos[0]} {worst_cos[1]:.9f}")
    assert not bad, (len(bad), bad[:20])
    for k, ref in st.P.items():       # step 1 of AdamW: +-lr; where |g| is well above rounding the parameters agree tightly
        d = (p_hip[k] - ref.detach()).abs()
        assert d.max().item() <= 2.2 * lr + 1e-7, (k, d.max().item())
        g = r["grads"][k]
        sure = g.abs() > 0.05 * r["grads"][k.replace("attn.key.bias", "attn.query.bias")].abs().max()
        if sure.any():
            assert d[sure].max().item() < 0.02 * lr, (k, d[sure].max().item())


def _flat_names(nm):
    from mebt_amd.engine import flat_layout
    return flat_layout(nm.n_layer, has_sos=nm.n_latent > 0)


def test_c2_bf16_logits_vs_oracle_b6():
    """forward at the benchmarked size and batch, eval and train(dropout 0): bf16 logits within a measured bound"""
    cfg = presets.sky_16f(dropout=0.0)
    ocfg = oracle_cfg_of(cfg)
    sd = perturbed_state(23, cfg)
    x, idx = batch(6, [4, 16, 16], 79)
    with torch.no_grad():
        ref, z_t, ntw, seq_len = orc.forward(sd, ocfg, x, idx, 0.5, training=True)
    m = presets.build_model(cfg, compute_dtype="bf16")
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    with torch.no_grad():
        logits, z2, _, _ = m(x.to(DEV), None, t=0.5, indices=idx.to(DEV))
    err = (logits.cpu() - ref).abs().max().item()
    print(f"[c2 bf16 B=6] max |dlogits| {err:.3e} (|logits| max {ref.abs().max().item():.2f})")
    from tests.helpers import record_measured
    record_measured("c2 bf16 B=6 logits: max |d| / max |logits|", err / ref.abs().max().item(), BF16_LOGITS_REL)
    assert torch.equal(z2.cpu(), z_t) and err < BF16_LOGITS_REL * ref.abs().max().item(), err


# ------------------------------------------------------------------------------------------------------------------
# C4: block 8192
# ------------------------------------------------------------------------------------------------------------------


@pytest.fixture(scope="module")
def ucf():
    cfg = presets.ucf_128f()
    sd = perturbed_state(31, cfg)
    return cfg, sd


def test_c4_revise_forward_vs_oracle(ucf):
    cfg, sd = ucf
    ocfg = oracle_cfg_of(cfg)
    B, N, NT = 1, 8192, 256
    g = torch.Generator().manual_seed(9)
    x = torch.randint(0, 16384, (B, 32, 16, 16), generator=g)
    perm = torch.randperm(N, generator=g)
    ci, ti = perm[:N - NT].unsqueeze(0), perm[N - NT:].unsqueeze(0)
    with torch.no_grad():
        ref = orc.reconstruct_mask(sd, ocfg, x.reshape(B, -1), ci, ti)
    # fp32: the north-star 1e-3 (measured 6.1e-6); bf16: measured 3.0e-2 abs = 0.9 % of max |logits|
    for dtype, tol in (("f32", 1e-3), ("bf16", BF16_LOGITS_REL * ref.abs().max().item())):
        m = presets.build_model(cfg, compute_dtype=dtype)
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        got, _ = m.reconstruct_mask(x.to(DEV), ci.to(DEV), ti.to(DEV))
        err = (got.cpu() - ref).abs().max().item()
        agree = (got.cpu().argmax(-1) == ref.argmax(-1)).float().mean().item()
        print(f"[c4 {dtype} NC=7936 NT=256] max |dlogits| {err:.3e}, arg-max agreement {agree:.4f}")
        from tests.helpers import record_measured
        record_measured(f"c4 revise forward {dtype} (7936, 256): max |dlogits|", err, tol, f"arg-max agreement {agree:.4f}")
        assert err < tol, (dtype, err)
        assert agree > (0.999 if dtype == "f32" else 0.95)       # bf16 measured 0.984
        del m
        torch.cuda.empty_cache()


def test_c4_cached_bf16_forward_vs_oracle(ucf):
    """The forward the config-4 benchmark legs TIME: bf16 engine + key / value cache of the latent_enc blocks
    (`mebt_forward_kvcache`) + gathered-key attention at 7936 keys (`AttnParams::kidx`, attn_fwd_pp) + bf16 logits, against
    `oracle.reconstruct_mask` (reference transformer.py:288-324, gpt.py:187-192) at block 8192, B = 1:
      1. first forward of a session, every context position re-projected, (NC, NT) = (7936, 256);
      2. the 256 targets re-sampled (new token ids), next revise chunk as targets, ONLY the 256 changed positions re-projected
         (`dirty` = the last 256 context columns, as `_gibbs_pass` names them) - the other 7680 key / value rows are the cache's;
      3. a third forward through `_KvSession`'s default (dirty = the previous forward's targets).
    Each against the oracle on the same ids / index sets at the bf16 gate (1.7 % of max |logits|; arg-max agreement > 0.95), with
    fp32 and with bf16 logits; and the production draw on those very logits against the oracle's inverse-CDF twin."""
    from mebt_amd.transformer import _KvSession
    from tests.helpers import record_measured
    cfg, sd = ucf
    ocfg = oracle_cfg_of(cfg)
    B, N, W = 1, 8192, 256
    g = torch.Generator().manual_seed(19)
    x = torch.randint(0, 16384, (B, N), generator=g)
    perm = torch.randperm(N, generator=g)
    chunk = [perm[k * W:(k + 1) * W] for k in range(N // W)]
    m = presets.build_model(cfg, compute_dtype="bf16")
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    nm = m._ensure_native()
    os.environ["MEBT_KV_CACHE_CHECK"] = "1"
    try:
        sess = {lp: _KvSession(nm, B, N) for lp in (False, True)}
    finally:
        del os.environ["MEBT_KV_CACHE_CHECK"]
    xs = x.clone()
    for step in range(3):
        # revise pass layout (mask_sampler.py:340-356): targets = chunk `step`, context = the other chunks with the chunk just re-sampled LAST
        ti = chunk[step].unsqueeze(0)
        rest = [chunk[k] for k in range(N // W) if k != step and k != step - 1]
        ci = torch.cat(rest + ([chunk[step - 1]] if step > 0 else [])).unsqueeze(0)
        assert ci.shape[1] == N - W
        dirty = None if step != 1 else ci[:, -W:].to(DEV)          # step 1: named by the loop; step 2: the session's default
        with torch.no_grad():
            ref = orc.reconstruct_mask(sd, ocfg, xs, ci, ti)
        tol = BF16_LOGITS_REL * ref.abs().max().item()
        for lp in (False, True):
            got = sess[lp].forward(xs.to(DEV), ci.to(DEV), ti.to(DEV), dirty, lp)
            assert got.dtype == (torch.bfloat16 if lp else torch.float32)
            gf = got.float().cpu()
            err = (gf - ref).abs().max().item()
            agree = (gf.argmax(-1) == ref.argmax(-1)).float().mean().item()
            print(f"[c4 bf16 cached forward {step} (7936, 256), {'bf16' if lp else 'fp32'} logits] max |dlogits| {err:.3e} (gate {tol:.3e}), arg-max agreement {agree:.4f}")
            record_measured(f"c4 cached bf16 forward {step} ({'bf16' if lp else 'fp32'} logits; {'all' if step == 0 else 256} rows re-projected): max |dlogits|",
                            err, tol, f"arg-max agreement {agree:.4f}")
            assert err < tol, (step, lp, err, tol)
            assert agree > 0.95, (step, lp, agree)
            if lp:      # the production draw (inverse CDF, one uniform per row) on the logits the loop would hand it, against its oracle twin
                # top-k keeps everything >= the k-th largest value (transformer.py:891-895); 8-bit logits tie there more often
                # than fp32 ones - how many candidates does `--top_k 32` really keep on bf16 logits?  (measured, recorded, bounded)
                kth = gf.view(W, -1).topk(32).values[:, -1:]
                kept = (gf.view(W, -1) >= kth).sum(1).float()
                kept_ref = (ref.view(W, -1) >= ref.view(W, -1).topk(32).values[:, -1:]).sum(1).float()
                record_measured(f"c4 cached bf16 forward {step}: candidates kept by top_k = 32 on bf16 logits, mean (max {int(kept.max())}; fp32 oracle logits: "
                                f"mean {kept_ref.mean().item():.2f} max {int(kept_ref.max())})", kept.mean().item(), 40.0)
                assert kept.mean().item() < 40.0 and kept.min().item() >= 32
                seed = 0x6A5D_0000_0100 + step
                for temp, k in ((1.0, 0), (1.0, 32)):
                    ids = torch.empty(B, W, dtype=torch.long, device=DEV)
                    _lib.check(_lib.load().mebt_op_sample_lp(_lib.ptr(got), 1, None, seed, temp, k, _lib.ptr(ids), None, None, None, B, W, W, 16384, 1,
                                                             _lib.cur_stream()))
                    torch.cuda.synchronize()
                    oid, _, margin = orc.sample_inverse_cdf(gf.view(W, -1), temp, k or None, seed)
                    diff = (ids.cpu().view(-1) != oid).nonzero().flatten().tolist()
                    assert len(diff) <= 2 and all(float(margin[r]) < 2e-5 for r in diff), (step, temp, k, diff, [float(margin[r]) for r in diff])
        sp, su = sess[True].rows_projected, sess[True].rows_uncached
        assert sp == (N - W) + step * W and su == (step + 1) * (N - W), (step, sp, su)
        # the loop's scatter: new ids at the targets just predicted
        xs = xs.clone()
        xs[0, chunk[step]] = torch.randint(0, 16384, (W,), generator=g)
    del m, sess
    torch.cuda.empty_cache()


def test_c4_bf16_draft_and_revise_every_forward_vs_oracle(ucf):
    """One whole `draft_and_revise` of the BENCHMARKED engine at block 8192 (bf16, key / value cache, gathered keys, bf16 logits,
    injected noise; draft 2 steps: NT = 8192, 4096; revise 8 steps: (7168, 1024)), B = 1 - every forward the loop runs against
    `oracle.reconstruct_mask` on the SAME partial state (ids, context set, target set as the HIP loop had them), each within the
    bf16 gate; and the ids the loop scattered = the oracle's draw (reference transformer.py:843-889) on the loop's own logits and
    noise, bit-exact except proven key ties.  (The fp32 engine's whole-loop identity: the next test.)"""
    from tests.helpers import record_measured
    cfg, sd = ucf
    ocfg = oracle_cfg_of(cfg)
    B, N = 1, 8192
    n_draft, n_revise, M = 2, 8, 1

    def stream(k, kind, shape):
        # fresh Exp(1) draws for every [1, NT, 16384] tensor (335 M values over the loop, ~3 s): the rolled-table noise of the fp32 tests
        # repeats every value 8 times per row, and bf16 logits repeat too (8 significant bits) - equal p with equal q is an EXACT key tie,
        # which the reference's unstable descending sort and the kernel (lowest index) break differently (first GPU run: 40 of 20 480 draws)
        g = torch.Generator().manual_seed(515100 + k)
        if kind == "perm":
            return torch.randperm(int(shape[0]), generator=g)
        return torch.empty(tuple(shape), dtype=torch.float32).exponential_(generator=g)

    m = presets.build_model(cfg, compute_dtype="bf16")
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    ctr = {"k": 0}
    recs = []

    def hook(kind, shape):
        k = ctr["k"]
        ctr["k"] += 1
        if kind == "exp" and len(shape) == 3:
            recs[-1]["noise_k"] = k
        return stream(k, kind, shape)

    m.noise_hook = hook
    orig = m._sampling_logits

    def rec(partial, c, t, *a, **kw):
        lg = orig(partial, c, t, *a, **kw)
        recs.append({"partial": partial.detach().cpu().clone(), "c": c.detach().cpu().clone().view(B, -1), "t": t.detach().cpu().clone().view(B, -1),
                     "logits": lg.detach().float().cpu(), "lp": lg.dtype == torch.bfloat16})
        return lg

    m._sampling_logits = rec
    x0 = torch.zeros(B, 32, 16, 16, dtype=torch.long)
    got = m.draft_and_revise(x0.to(DEV), None, n_draft, 1.0, None, None, n_revise, 1.0, None, None, M, False).cpu()
    torch.cuda.synchronize()
    assert len(recs) == n_draft + M * n_revise and all(r["lp"] for r in recs)          # bf16 logits at temperature 1.0
    assert recs[0]["c"].shape[1] == 0 and recs[0]["t"].shape[1] == 8192 and recs[-1]["c"].shape[1] == 7168
    proj, unc = m._kv_last
    assert proj < unc, (proj, unc)            # the cache was on: fewer context rows projected than the forwards read
    worst, n_tie = 0.0, 0
    for i, r in enumerate(recs):
        with torch.no_grad():
            ref = orc.reconstruct_mask(sd, ocfg, r["partial"].view(B, -1), r["c"], r["t"])
        mx = ref.abs().max().item()
        err = (r["logits"] - ref).abs().max().item()
        agree = (r["logits"].argmax(-1) == ref.argmax(-1)).float().mean().item()
        worst = max(worst, err / mx)
        print(f"[c4 bf16 draft_and_revise forward {i}: NC {r['c'].shape[1]} NT {r['t'].shape[1]}] max |dlogits| {err:.3e} = {100 * err / mx:.2f} % of max, arg-max agreement {agree:.4f}")
        assert err < BF16_LOGITS_REL * mx, (i, err, mx)
        assert agree > 0.95, (i, agree)
        # the ids the loop wrote at this forward's targets = the reference's draw on the loop's logits and noise
        nxt = recs[i + 1]["partial"].view(B, -1) if i + 1 < len(recs) else got.view(B, -1)
        wrote = nxt.gather(1, r["t"])
        nz = stream(r["noise_k"], "exp", tuple(r["logits"].shape))
        oid, oprobs = orc.sample_from_logits(r["logits"], 1.0, None, None, nz)
        for b, j in (wrote != oid).nonzero().tolist():
            top2 = (oprobs[b, j].double() / nz[b, j].double()).topk(2).values
            assert top2[0] / top2[1] < 1 + 5e-4, ("not a tie", i, b, j, float(top2[0] / top2[1]))
            n_tie += 1
        del ref
    assert n_tie <= 8, n_tie
    record_measured("c4 bf16 draft_and_revise (2 + 8 forwards, cached, bf16 logits): worst max |dlogits| / max |logits| over the forwards", worst,
                    BF16_LOGITS_REL, f"{n_tie} proven draw ties; context rows projected {proj} of {unc}")


def test_c4_draft_and_revise_block8192_bit_exact(ucf):
    """A whole draft_and_revise at block 8192 (draft 2 steps: NT = 8192, 4096; revise 8 steps: NC = 7168, NT = 1024),
    B = 1, fp32 parity mode, against oracle.draft_and_revise driven by the same permutations and Exp(1) noise.
    Token ids must be identical; a mismatch is tolerated only where the ORACLE's own two best keys p/q are within
    5e-4 of each other (an fp tie that 1e-5 logit differences may flip), which the step-by-step replay below proves."""
    cfg, sd = ucf
    ocfg = oracle_cfg_of(cfg)
    B, N = 1, 8192
    n_draft, n_revise, M = 2, 8, 1

    def stream(k, kind, shape):
        g = torch.Generator().manual_seed(424200 + k)
        if kind == "perm":
            return torch.randperm(int(shape[0]), generator=g)
        return torch.empty(shape, dtype=torch.float32).exponential_(generator=g)

    # ---- product (HIP) end to end
    m = presets.build_model(cfg, compute_dtype="f32")
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    ctr = {"k": 0}

    def hook(kind, shape):
        k = ctr["k"]
        ctr["k"] += 1
        return stream(k, kind, shape)

    m.noise_hook = hook
    x0 = torch.zeros(B, 32, 16, 16, dtype=torch.long)
    got = m.draft_and_revise(x0.to(DEV), None, n_draft, 1.0, None, None, n_revise, 1.0, None, None, M, False).cpu()
    n_draws = ctr["k"]

    # ---- oracle with the same streams, recording every step
    octr = {"k": 0}
    steps = []

    def perm_fn(tag, B_, N_):
        out = []
        for _ in range(B_):
            out.append(stream(octr["k"], "perm", (N_,)))
            octr["k"] += 1
        return torch.stack(out)

    def noise_fn(tag, shape):
        k = octr["k"]
        octr["k"] += 1
        nz = stream(k, "exp", tuple(shape))
        steps[-1]["noise_k"] = k
        return nz

    def logits_fn(partial, c, t_):
        with torch.no_grad():
            lg = orc.reconstruct_mask(sd, ocfg, partial, c, t_)
        steps.append({"partial": partial.clone(), "c": c.clone(), "t": t_.clone(), "logits": lg})
        return lg

    with torch.no_grad():
        ref = orc.draft_and_revise(sd, ocfg, x0, n_draft, 1.0, None, None, n_revise, 1.0, None, None, M, False, perm_fn,
                                   noise_fn, logits_fn=logits_fn)
    assert octr["k"] == n_draws == B * (1 + M) + n_draft + M * n_revise
    assert len(steps) == n_draft + M * n_revise
    assert steps[0]["c"].shape[1] == 0 and steps[0]["t"].shape[1] == 8192 and steps[-1]["c"].shape[1] == 7168
    if torch.equal(got, ref):
        return
    # ---- replay every step on the HIP path from the ORACLE's state: ids may differ only at provable fp ties
    from mebt_amd.transformer import sample_from_logits_scored
    n_tie = 0
    for s in steps:
        lg, _ = m.reconstruct_mask(s["partial"].to(DEV), s["c"].to(DEV), s["t"].to(DEV))
        assert (lg.cpu() - s["logits"]).abs().max().item() < 1e-3
        nz = stream(s["noise_k"], "exp", tuple(s["logits"].shape))
        ids, _, _ = sample_from_logits_scored(lg, 1.0, None, None, nz.to(DEV))
        oid, _ = orc.sample_from_logits(s["logits"], 1.0, None, None, nz)
        diff = (ids.cpu() != oid).nonzero()
        for b, j in diff.tolist():
            p = torch.softmax(s["logits"][b, j].double(), -1)
            key = p / nz[b, j].double()
            top2 = key.topk(2).values
            assert top2[0] / top2[1] < 1 + 5e-4, ("not a tie", b, j, float(top2[0] / top2[1]))
            n_tie += 1
    assert 0 < n_tie <= 3, n_tie        # got != ref must be explained by at least one flipped tie, and ties are rare


def _grad_report(g_hip, grads_ref, grad_gate, cos_gate, rell2_gate):
    """every gradient tensor against the oracle's: max error relative to the tensor's max, cosine, relative L2"""
    worst, worst_cos, worst_l2, bad = ("", 0.0), ("", 1.0), ("", 0.0), []
    for k, ref in grads_ref.items():
        denom = grads_ref[k.replace("attn.key.bias", "attn.query.bias")].abs().max().item() + 1e-12
        err = (g_hip[k] - ref).abs().max().item() / denom
        worst = max(worst, (k, err), key=lambda v: v[1])
        if not err < grad_gate:
            bad.append((k, round(err, 5)))
        if "attn.key.bias" in k:
            continue                                  # a mathematically zero gradient has no direction
        a, b = g_hip[k].double().reshape(-1), ref.double().reshape(-1)
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        rl2 = float((a - b).norm() / (b.norm() + 1e-300))
        worst_cos = min(worst_cos, (k, cos), key=lambda v: v[1])
        worst_l2 = max(worst_l2, (k, rl2), key=lambda v: v[1])
        if not (cos >= cos_gate and rl2 <= rell2_gate):
            bad.append((k, "cos", round(cos, 7), "rel-L2", round(rl2, 6)))
    return worst, worst_cos, worst_l2, bad


@pytest.mark.parametrize("dtype,window", [pytest.param("f32", None, marks=long_only), ("bf16", None), pytest.param("bf16", (16, 9), marks=long_only),
                                          pytest.param("f32", (16, 9), marks=long_only)])
def test_c4_train_step_vs_oracle(ucf, dtype, window):
    """TRAINING at the 128-frame geometry (VERDICT r03 missing #3 / weak #8: the HIP backward had never run with 4096+ keys or a
    weight-gradient reduction over 8192+ tokens).  UCF-128f preset (configs/ucf/mebt_128f.yaml:4-57: block 8192, grid [32,16,16],
    budget 8192, t_prior gaussian100000_2), B = 2, one whole `TrainLoop.step` against `oracle.train_step`: loss, every gradient,
    post-step parameters.  window = None: the full sequence (T = 32 drawn: NC = NT = 4096 at t = 0.5, i.e. latent_enc over 4096 keys,
    lt2l over 4352 keys, latent_dec with 4096 queries, head over 8192 rows); window = (16, 9): the mask sampler's two numpy draws
    (mask_sampler.py:88,90) forced to T = 16 latent frames starting at frame 9, so seq_len = 4096 < N = 8192 and the loss divisor
    and the positional rows follow the window (transformer.py:243-259, mask_sampler.py:83-99).  fp32 engine: north-star 1e-3;
    bf16 engine: max error on the C2 bound, cosine / relative L2 on bounds measured at this geometry (C4_BF16_*)."""
    cfg, sd = ucf
    assert cfg.model.params.t_prior == "gaussian100000_2" and cfg.model.mask.params.budget == 8192
    lr = cfg.exp.exact_lr
    ocfg = oracle_cfg_of(cfg)
    B, t = 2, 0.5
    x, idx = batch(B, [32, 16, 16], 91)
    m = presets.build_model(cfg, compute_dtype=dtype)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    loop = TrainLoop(m, fused_optimizer=False)
    rec = {}
    real_choice, real_randint = np.random.choice, np.random.randint
    T_, start_ = window if window is not None else (32, 0)

    def fake_choice(a, p=None, **kw):
        rec["a"], rec["p"] = np.asarray(a).copy(), np.asarray(p).copy()
        return T_

    def fake_randint(lo, hi=None, **kw):
        rec["randint"] = (lo, hi)
        return start_
    np.random.choice, np.random.randint = fake_choice, fake_randint
    try:
        stats = loop.step(x.to(DEV), idx.to(DEV), t=t).cpu()
    finally:
        np.random.choice, np.random.randint = real_choice, real_randint
    assert list(rec["a"]) == list(range(1, 33)) and abs(rec["p"].sum() - 1) < 1e-9      # the prior over 1..32 latent frames
    if window is not None:
        assert tuple(rec["randint"]) == (0, 32 - T_ + 1)
    seq_len = T_ * 256
    nm = loop.native
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    g_hip = {k: v.detach().cpu().clone() for k, v in nm.views(shapes, grads=True).items()}
    p_hip = {k: v.detach().cpu().clone() for k, v in nm.views(shapes).items()}
    assert m.weight_decay == cfg.exp.weight_decay == 1e-4
    st = orc.TrainState(sd, lr=lr, weight_decay=cfg.exp.weight_decay)
    r = orc.train_step(st, ocfg, x, idx, t, window=(T_, start_) if window is not None else None)
    assert r["n_targets"] == int(stats[3]) == B * seq_len // 2
    loss_hip = float(stats[4])
    f32 = dtype == "f32"
    assert abs(loss_hip - r["loss"]) < (2e-5 if f32 else C2_BF16_LOSS_REL) * abs(r["loss"]), (loss_hip, r["loss"])
    worst, worst_cos, worst_l2, bad = _grad_report(g_hip, r["grads"], C2_F32_GRAD if f32 else C2_BF16_GRAD, (1 - 1e-6) if f32 else C4_BF16_COS,
                                                   1e-3 if f32 else C4_BF16_RELL2)
    tag = f"c4 train {dtype} window={window}"
    print(f"[{tag}] seq_len {seq_len}: loss hip {loss_hip:.6f} oracle {r['loss']:.6f}; worst gradient {worst[0]} rel-to-max {worst[1]:.3e}; "
          f"worst cosine {worst_cos[0]} {worst_cos[1]:.8f}; worst rel-L2 {worst_l2[0]} {worst_l2[1]:.3e}")
    from tests.helpers import record_measured
    record_measured(f"{tag}: worst gradient rel-to-max", worst[1], C2_F32_GRAD if f32 else C2_BF16_GRAD, worst[0])
    record_measured(f"{tag}: worst 1-cosine", 1 - worst_cos[1], 1e-6 if f32 else 1 - C4_BF16_COS, worst_cos[0])
    record_measured(f"{tag}: worst rel-L2", worst_l2[1], 1e-3 if f32 else C4_BF16_RELL2, worst_l2[0])
    assert not bad, (len(bad), bad[:20])
    for k, ref in st.P.items():       # step 1 of AdamW moves a parameter by ~lr sign(g): tight where |g| is well above rounding
        d = (p_hip[k] - ref.detach()).abs()
        assert d.max().item() <= 2.2 * lr + 1e-7, (k, d.max().item())
        g = r["grads"][k]
        sure = g.abs() > (0.05 if f32 else 0.25) * r["grads"][k.replace("attn.key.bias", "attn.query.bias")].abs().max()
        if sure.any():
            assert d[sure].max().item() < (0.02 if f32 else 0.12) * lr, (k, d[sure].max().item())


@pytest.mark.parametrize("B,t", [(6, 0.0), pytest.param(6, 0.03, marks=long_only), (6, 0.97), (5, 0.337), pytest.param(3, 0.62, marks=long_only), (1, 0.81),
                                 pytest.param(7, 0.25, marks=long_only)])
def test_c2_ragged_shapes_bf16_engine_vs_fp32_engine(B, t):
    """The benchmarked shapes are the friendliest ones (B = 6, NC = NT = 512: every GEMM dimension a multiple of every tile).
    Real training draws t ~ U(0,1) and the last batch of an epoch is short, so the tuned bf16 kernels also see
    NT = 1024 / NC = 0, NT = 31 / NC = 993, 5 x 256 = 1280 latent rows, one sample, seven samples ...: ragged tile edges in
    M (forward, dgrad) and in K (weight gradients reduce over the token count), buffer-bounds zero fill, other tuner
    buckets.  One full-width train step in bf16 against the exact-fp32 engine on the same weights / batch / t: loss and
    every gradient, on the measured C2 bounds."""
    cfg = presets.sky_16f(dropout=0.0)
    sd = perturbed_state(21, cfg)
    x, idx = batch(B, [4, 16, 16], 100 + B)
    shapes = {k: tuple(v.shape) for k, v in sd.items()}

    def run(dtype):
        m = presets.build_model(cfg, compute_dtype=dtype)
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        loop = TrainLoop(m, fused_optimizer=False)
        stats = loop.step(x.to(DEV), idx.to(DEV), t=t).cpu()
        grads = {k: v.detach().cpu().clone() for k, v in loop.native.views(shapes, grads=True).items()}
        del loop, m
        torch.cuda.empty_cache()
        return stats, grads

    s32, g32 = run("f32")
    s16, g16 = run("bf16")
    assert int(s32[3]) == int(s16[3]) and int(s32[3]) > 0
    assert abs(float(s16[4]) - float(s32[4])) < C2_BF16_LOSS_REL * abs(float(s32[4])), (float(s16[4]), float(s32[4]))
    worst, bad = ("", 0.0), []
    for k, ref in g32.items():
        denom = g32[k.replace("attn.key.bias", "attn.query.bias")].abs().max().item() + 1e-12
        if denom < 1e-10:            # nothing reaches this parameter (tok_emb / latent_enc keys with NC = 0): both must be zero
            assert g16[k].abs().max().item() == 0.0 and ref.abs().max().item() == 0.0, k
            continue
        err = (g16[k] - ref).abs().max().item() / denom
        if err > worst[1]:
            worst = (k, err)
        if not err < C2_BF16_GRAD:
            bad.append((k, round(err, 4)))
    print(f"[c2 ragged B={B} t={t}] NT/sample {int(s32[3]) // B}: loss {float(s16[4]):.5f} vs {float(s32[4]):.5f}; worst gradient {worst[0]} {worst[1]:.3e}")
    assert not bad, (len(bad), bad[:20])


def _replay_steps_prove_ties(m, steps, stream, temperature, max_ties=3, top_k=None, min_ties=1, tie_steps=None):
    """`got != ref` after a sampling loop: replay every recorded step of the ORACLE on the HIP path from the oracle's own state —
    logits within 1e-3, and sampled ids different only where the oracle's own numbers say a logit difference of the measured
    size may flip the draw, after which the two runs legitimately diverge:
      * the oracle's two best keys p / q are within 5e-4 (an fp tie), or
      * with top-k: the oracle's k-th and (k+1)-th logits of that row are closer than twice this step's measured
        max |logits_hip - logits_oracle|, so the two paths keep different 32-sets and one of the two draws is the element
        the other dropped (torch.topk boundary, modules/gpt.py:233-238)."""
    from mebt_amd.transformer import sample_from_logits_scored
    n_tie = 0
    for si, s in enumerate(steps):
        n_before = n_tie
        lg, _ = m.reconstruct_mask(s["partial"].to(DEV), s["c"].to(DEV), s["t"].to(DEV))
        d_lg = (lg.cpu() - s["logits"]).abs().max().item()
        assert d_lg < 1e-3
        nz = stream(s["noise_k"], "exp", tuple(s["logits"].shape))
        T_, k_ = (1.0, None) if s.get("boot") else (temperature, top_k)       # the bootstrap phase samples at T = 1 without top-k (:41-42)
        ids, _, _ = sample_from_logits_scored(lg, T_, k_, None, nz.to(DEV))
        oid, oprobs = orc.sample_from_logits(s["logits"], T_, k_, None, nz)
        ids = ids.cpu()
        for b, j in (ids != oid).nonzero().tolist():
            top2 = (oprobs[b, j].double() / nz[b, j].double()).topk(2).values     # the oracle's own keys p / q after temperature / top-k
            if top2[0] / top2[1] < 1 + 5e-4:
                n_tie += 1
                continue
            assert k_, ("not a tie", b, j, float(top2[0] / top2[1]))
            kth = s["logits"][b, j].topk(k_ + 1).values
            gap = float(kth[k_ - 1] - kth[k_])
            edge = {int(i) for i in (s["logits"][b, j] >= kth[k_] - 2 * d_lg).nonzero().flatten().tolist()
                    if float(s["logits"][b, j, i]) <= float(kth[k_ - 1]) + 2 * d_lg}       # elements either path may put on the other side of the cut
            assert gap <= 2 * d_lg and (int(ids[b, j]) in edge or int(oid[b, j]) in edge), \
                ("neither a key tie nor a top-k boundary", b, j, float(top2[0] / top2[1]), gap, d_lg)
            n_tie += 1
        if tie_steps is not None and n_tie > n_before:
            tie_steps.append(si)                         # the forwards whose draw has a proven tie: the runs may part ways right behind them
    assert min_ties <= n_tie <= max_ties, n_tie        # a difference must be explained by at least one flipped tie, and ties are rare
    return n_tie


class _MaskOrderRecorder:
    """Records every `gumbel_top_k` call of an oracle sampling run (mask_sampler.py:178-187: the confidence keys
    score / sum / noise^ctemp whose descending order picks the new context tokens and ORDERS the remaining targets), keyed by the
    forward step that preceded it.  Two targets whose keys agree to ~1e-5 can come out in the other order on the HIP path (its
    scores carry the 7e-6 logit difference); the order of the targets decides which noise row meets which position, so the two
    runs legitimately diverge from there — the third kind of fp tie of a sampling loop, after the draw's key tie and the top-k cut."""

    def __init__(self, steps):
        self.steps, self.rec, self._orig = steps, {}, None

    def __enter__(self):
        self._orig = orc.gumbel_top_k

        def patched(score, ctemp, noise):
            order = self._orig(score, ctemp, noise)
            prob = score / score.sum(-1, keepdim=True)
            self.rec[len(self.steps) - 1] = ((prob / (noise ** ctemp)).clone(), order.clone())
            return order
        orc.gumbel_top_k = patched
        return self

    def __exit__(self, *a):
        orc.gumbel_top_k = self._orig

    def prove_order_ties(self, hsteps, max_ties=64, rel=2e-4, draw_tie_steps=()):
        """`hsteps`: the (c, t, partial) the HIP run really fed to its forwards, in order.  Finds the first forward whose inputs differ
        from the oracle's and proves that difference is a re-ordering of near-equal confidence keys: same context SET, same target
        SET, same token ids so far, and at every position where the target order differs the two elements' oracle keys are within
        `rel` of each other.  Returns (first differing step, number of swapped positions); (None, 0) if the inputs never differ."""
        for i, (h, o) in enumerate(zip(hsteps, self.steps)):
            same = h["c"].shape == o["c"].shape and h["t"].shape == o["t"].shape and torch.equal(h["c"], o["c"]) and torch.equal(h["t"], o["t"]) \
                and torch.equal(h["partial"].reshape(-1), o["partial"].reshape(-1))
            if same:
                continue
            if i > 0 and (i - 1) in draw_tie_steps and not torch.equal(h["partial"].reshape(-1), o["partial"].reshape(-1)):
                return i, 0          # the runs part ways at a PROVEN draw tie of forward i - 1 (step-by-step replay): nothing to explain here
            assert i > 0 and (i - 1) in self.rec, ("inputs differ at a step no mask re-ordering precedes", i)
            assert h["c"].shape == o["c"].shape and h["t"].shape == o["t"].shape, (i, h["c"].shape, o["c"].shape)
            assert torch.equal(h["partial"].reshape(-1), o["partial"].reshape(-1)), ("token ids differ before any re-ordering", i)
            keys, _ = self.rec[i - 1]
            prev_t = self.steps[i - 1]["t"]
            n_swapped = 0
            nc_old = self.steps[i - 1]["c"].shape[1]
            for b in range(h["t"].shape[0]):
                # the key order = [new context tokens ..., remaining targets ...]: compare position by position; the old context is untouched
                assert torch.equal(h["c"][b, :nc_old], o["c"][b, :nc_old]), ("old context differs", i, b)
                hh = torch.cat([h["c"][b, nc_old:], h["t"][b]])
                oo = torch.cat([o["c"][b, nc_old:], o["t"][b]])
                assert sorted(hh.tolist()) == sorted(oo.tolist()), ("not a permutation of the same tokens", i, b)
                key_of = dict(zip(prev_t[b].tolist(), keys[b].double().tolist()))
                for j in (hh != oo).nonzero().flatten().tolist():
                    ka, kb = key_of[int(hh[j])], key_of[int(oo[j])]
                    assert abs(ka - kb) <= rel * max(abs(ka), abs(kb)), ("re-ordered elements are not a key tie", i, b, j, ka, kb)
                    n_swapped += 1
            assert 0 < n_swapped <= max_ties, n_swapped
            return i, n_swapped
        return None, 0


def _record_hip_forwards(m):
    """wrap m.reconstruct_mask: the (partial, c, t) of every forward the HIP sampling loop runs"""
    hsteps, orig = [], m.reconstruct_mask

    def rec(partial, c, t, *a, **kw):
        hsteps.append({"partial": partial.detach().cpu().clone(), "c": c.detach().cpu().clone(), "t": t.detach().cpu().clone()})
        return orig(partial, c, t, *a, **kw)
    m.reconstruct_mask = rec
    return hsteps


def test_c4_bidirect_sample_bootstrap_topk_block8192(ucf):
    """The shipped UCF-128f sampling flow at its real geometry (VERDICT r03 missing #4 / weak #9; reference
    scripts/valid_dnr_config_ckpt_exp_ucf_128f.sh:10-15 -> sample_vqgan_transformer_videos.py:22-94): `bidirect_sample` of one
    128-frame clip (block 8192) with `--bootstrap B --top_k 32`, then `vid_n_steps` MaskGIT steps at context temperature 2.0,
    cosine mask schedule, fp32 engine, against `oracle.bidirect_sample` driven by the same noise: code map identical, or every
    difference a proven fp tie of the oracle (step-by-step replay from the oracle's state), score within 1e-4.  The shipped
    schedule (bootstrap 64 + 32 steps = 96 forwards of ~1.7 TFLOP on the host for the oracle) runs with MEBT_LONG_TESTS=1
    (result committed under profiles/); the default run uses bootstrap 2 + 3 steps: the same kernels and shapes (NT from
    8192 down, the [1, 8192, 16384] probability maps, top-k thresholds over 8192 rows)."""
    from mebt_amd.sampling import bidirect_sample
    cfg, sd = ucf
    ocfg = oracle_cfg_of(cfg)
    long_run = os.environ.get("MEBT_LONG_TESTS") == "1"
    boot, n_steps = (64, 32) if long_run else (2, 3)
    base = torch.empty(8192, 2048).exponential_(generator=torch.Generator().manual_seed(77001))

    def stream(k, kind, shape):
        g = torch.Generator().manual_seed(616100 + k)
        if kind == "perm":
            return torch.randperm(int(shape[0]), generator=g)
        if kind == "randn":
            return torch.randn(tuple(shape), generator=g)
        if len(shape) == 3 and shape[-1] == 16384:       # [1, NT, V] Exp(1): rolled copies of one table instead of 134 M fresh draws per step
            nt = shape[1]
            rows = base.roll(shifts=37 * k, dims=0)[:nt]
            return torch.cat([rows.roll(shifts=k * 131 + 17 * i, dims=1) for i in range(8)], dim=1).reshape(shape)
        return torch.empty(tuple(shape), dtype=torch.float32).exponential_(generator=g)

    m = presets.build_model(cfg, compute_dtype="f32")
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.mask_sampler.schedule = "cosine"                       # sample_vqgan_transformer_videos.py:189,219
    ctr = {"k": 0}

    def hook(kind, shape):
        k = ctr["k"]
        ctr["k"] += 1
        return stream(k, kind, shape)

    m.noise_hook = hook
    m.mask_sampler.noise_hook = hook
    hsteps = _record_hip_forwards(m)
    import time
    t0 = time.time()
    log = bidirect_sample(m, 1, 128, 128, 128, temperature=1.0, top_k=32, top_p=None, vid_n_steps=n_steps, vid_c_temp=2.0,
                          ctemp_schedule="linear", strategy="maskgit", bootstrap=boot)
    torch.cuda.synchronize()
    t_hip = time.time() - t0
    got = log["code_maps"].cpu()
    assert tuple(got.shape) == (1, 32, 16, 16)
    octr = {"k": 0}
    steps = []

    def noise_fn(tag, shape):
        k = octr["k"]
        octr["k"] += 1
        if tag == "sample":
            steps[-1]["noise_k"] = k
        if tag == "randn":
            steps[-1]["boot"] = True
        return stream(k, "randn" if tag == "randn" else "exp", tuple(shape))

    def logits_fn(partial, c_, t_):
        with torch.no_grad():
            lg = orc.reconstruct_mask(sd, ocfg, partial, c_, t_)
        steps.append({"partial": partial.clone(), "c": c_.clone(), "t": t_.clone(), "logits": lg})
        return lg

    import copy
    ocos = copy.copy(ocfg)
    ocos.schedule = "cosine"
    t0 = time.time()
    with torch.no_grad(), _MaskOrderRecorder(steps) as mask_rec:
        ref, score = orc.bidirect_sample(sd, ocos, 1, 128, 128, 128, 1.0, 32, None, n_steps, 2.0, noise_fn, ctemp_schedule="linear",
                                         strategy="maskgit", bootstrap=boot, logits_fn=logits_fn)
    t_orc = time.time() - t0
    assert octr["k"] == ctr["k"], (octr["k"], ctr["k"])          # both sides consumed the same draws in the same order
    assert sum(1 for s_ in steps if s_.get("boot")) == boot and len(steps) > boot + n_steps // 2      # `sample` skips steps whose context is bigger than expected (:400-402)
    assert steps[0]["c"].shape[1] == 0 and steps[0]["t"].shape[1] == 8192 and steps[boot]["c"].shape[1] == boot
    same = bool(torch.equal(got, ref))
    n_tie, n_swap, first = 0, 0, None
    if same:
        np.testing.assert_allclose(log["score"].cpu().numpy(), score.numpy(), rtol=1e-4)
    else:       # every forward replayed from the oracle's state (draw ties), then the first forward whose inputs differ (mask-order ties)
        m.reconstruct_mask = type(m).reconstruct_mask.__get__(m)
        tie_steps = []
        n_tie = _replay_steps_prove_ties(m, steps, stream, 1.0, max_ties=64, top_k=32, min_ties=0, tie_steps=tie_steps)
        first, n_swap = mask_rec.prove_order_ties(hsteps, draw_tie_steps=tie_steps)
        assert n_tie + n_swap > 0
    msg = (f"[c4 bidirect_sample block 8192 bootstrap {boot} top_k 32 steps {n_steps}] {len(steps)} forwards; code map == oracle: {same} "
           f"(proven fp ties in the replay: {n_tie} draws, {n_swap} re-ordered targets of near-equal confidence at forward {first}); HIP {t_hip:.1f} s, oracle {t_orc:.1f} s")
    print(msg)
    from tests.helpers import record_measured
    record_measured(msg, n_tie + n_swap, 128)



def _rolled_noise_stream(seed, rows):
    """noise for long sampling loops: `[B, NT, 16384]` Exp(1) tensors are rolled copies of one [rows, 2048] table (fresh draws of
    134 M values per step would dominate the test), everything else is drawn from a generator keyed by the call number"""
    base = torch.empty(rows, 2048).exponential_(generator=torch.Generator().manual_seed(seed))

    def stream(k, kind, shape):
        g = torch.Generator().manual_seed(seed * 8 + 100 + k)
        if kind == "perm":
            return torch.randperm(int(shape[0]), generator=g)
        if kind == "randn":
            return torch.randn(tuple(shape), generator=g)
        if len(shape) == 3 and shape[-1] == 16384:
            out = []
            for b in range(shape[0]):
                r = base.roll(shifts=37 * k + 211 * b, dims=0)[:shape[1]]
                out.append(torch.cat([r.roll(shifts=k * 131 + 17 * i + 7 * b, dims=1) for i in range(8)], dim=1))
            return torch.stack(out).reshape(shape)
        return torch.empty(tuple(shape), dtype=torch.float32).exponential_(generator=g)
    return stream


def _sky_sampling_model(seed):
    cfg = presets.sky_16f(dropout=0.0)
    sd = perturbed_state(seed, cfg)
    m = presets.build_model(cfg, compute_dtype="f32")
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.mask_sampler.schedule = "cosine"                       # sample_vqgan_transformer_videos.py:189,219
    import copy
    ocfg = copy.copy(oracle_cfg_of(cfg))
    ocfg.schedule = "cosine"
    return m, sd, ocfg


def _drive_both(m, stream, hip_call, oracle_call):
    """run the HIP driver and the oracle driver on the same stream of draws; returns (hip result, oracle result, recorded oracle steps)"""
    ctr, octr, steps = {"k": 0}, {"k": 0}, []

    def hook(kind, shape):
        k = ctr["k"]
        ctr["k"] += 1
        return stream(k, kind, shape)

    m.noise_hook = hook
    m.mask_sampler.noise_hook = hook
    hsteps = _record_hip_forwards(m)
    got = hip_call()
    torch.cuda.synchronize()
    m.reconstruct_mask = type(m).reconstruct_mask.__get__(m)

    def noise_fn(tag, shape):
        k = octr["k"]
        octr["k"] += 1
        if tag == "sample":
            steps[-1]["noise_k"] = k
        return stream(k, "randn" if tag == "randn" else "exp", tuple(shape))

    def logits_fn(partial, c_, t_):
        steps.append({"partial": partial.clone(), "c": c_.clone(), "t": t_.clone()})
        return None

    with _MaskOrderRecorder(steps) as mask_rec:
        ref = oracle_call(noise_fn, logits_fn, steps)
    assert octr["k"] == ctr["k"], (octr["k"], ctr["k"])          # both sides consumed the same draws in the same order
    return got, ref, steps, hsteps, mask_rec


def _explain_difference(m, steps, hsteps, mask_rec, stream, temperature, top_k=None):
    """code maps differ: draw ties (every oracle step replayed on the HIP path) and / or a re-ordering of near-equal confidences"""
    tie_steps = []
    n_tie = _replay_steps_prove_ties(m, steps, stream, temperature, max_ties=16, top_k=top_k, min_ties=0, tie_steps=tie_steps)
    first, n_swap = mask_rec.prove_order_ties(hsteps, draw_tie_steps=tie_steps)
    assert n_tie + n_swap > 0
    return n_tie, n_swap, first


def test_c2_bidirect_sample_sliding_window_continuation():
    """f1 at real geometry, the part the block-8192 test does not reach (VERDICT r04 missing #3): `bidirect_sample` with
    `total_length` > one window — Sky-16f geometry (block 1024 = 4 latent frames of 16 x 16), 32 video frames from 16-frame
    windows with 8 frames of fixed context, i.e. a first window from nothing and TWO continuations whose context index set is the
    fixed first two latent frames (reference sample_vqgan_transformer_videos.py:55-71) — fp32 engine against
    `oracle.bidirect_sample` on the same noise: code map [1, 8, 16, 16] identical, or every difference a proven fp tie of the
    oracle (draw ties: step-by-step replay from the oracle's own state; confidence-order ties: `_MaskOrderRecorder`); score of the
    first window within 1e-4.  (Batch 2 runs in `test_c2_extrapolate_edit_mode`.)"""
    from mebt_amd.sampling import bidirect_sample
    m, sd, ocfg = _sky_sampling_model(41)
    stream = _rolled_noise_stream(88001, 1024)
    n_steps = 4

    def oracle_call(noise_fn, logits_fn, steps):
        def lf(partial, c_, t_):
            logits_fn(partial, c_, t_)
            with torch.no_grad():
                steps[-1]["logits"] = orc.reconstruct_mask(sd, ocfg, partial, c_, t_)
            return steps[-1]["logits"]
        with torch.no_grad():
            return orc.bidirect_sample(sd, ocfg, 1, 32, 16, 8, 1.0, None, None, n_steps, 2.0, noise_fn, logits_fn=lf)

    log, (ref, score), steps, hsteps, mask_rec = _drive_both(
        m, stream, lambda: bidirect_sample(m, 1, 32, 16, 8, temperature=1.0, top_k=None, top_p=None, vid_n_steps=n_steps, vid_c_temp=2.0),
        oracle_call)
    got = log["code_maps"].cpu()
    assert tuple(got.shape) == (1, 8, 16, 16) and tuple(ref.shape) == (1, 8, 16, 16)
    # the continuation windows really ran with the fixed context: 512 context indices 0..511 at their first step
    cont = [s_ for s_ in steps if s_["c"].shape[1] >= 512 and torch.equal(s_["c"][0, :512], torch.arange(512))]
    assert len(cont) >= 2 and steps[0]["c"].shape[1] == 0 and steps[0]["t"].shape[1] == 1024
    same = bool(torch.equal(got, ref))
    why = None
    if same:
        np.testing.assert_allclose(log["score"].cpu().numpy(), score.numpy(), rtol=1e-4)
    else:
        why = _explain_difference(m, steps, hsteps, mask_rec, stream, 1.0)
    print(f"[c2 bidirect_sample 32 frames from 16-frame windows] {len(steps)} forwards, {len(cont)} of them continuations; code map == oracle: {same} (proven ties: {why})")


def test_c2_extrapolate_edit_mode():
    """`extrapolate` (reference sample_vqgan_transformer_videos.py:96-157) at Sky-16f geometry: a given [2, 4, 16, 16] code map
    continued to 32 video frames (8 latent frames) in two jumps of `sample(..., edit=True)` — the mask schedule then counts the
    512 edited positions instead of the block (transformer.py:373-376,399) — fp32 engine against `oracle.extrapolate` on the same
    noise, with top-k 64: ids identical, or every difference a proven tie / top-k boundary of the oracle."""
    from mebt_amd.sampling import extrapolate
    m, sd, ocfg = _sky_sampling_model(43)
    stream = _rolled_noise_stream(88003, 512)
    vq0 = torch.randint(0, 16384, (2, 4, 16, 16), generator=torch.Generator().manual_seed(9))
    n_steps = 4

    def oracle_call(noise_fn, logits_fn, steps):
        def lf(partial, c_, t_):
            logits_fn(partial, c_, t_)
            with torch.no_grad():
                steps[-1]["logits"] = orc.reconstruct_mask(sd, ocfg, partial, c_, t_)
            return steps[-1]["logits"]
        with torch.no_grad():
            return orc.extrapolate(sd, ocfg, vq0, 32, 16, 8, 0.9, 64, None, n_steps, 2.5, noise_fn, logits_fn=lf)

    log, ref, steps, hsteps, mask_rec = _drive_both(
        m, stream, lambda: extrapolate(m, vq0.to(DEV), 32, 16, 8, temperature=0.9, top_k=64, top_p=None, vid_n_steps=n_steps, vid_c_temp=2.5),
        oracle_call)
    got = log["code_maps"].cpu()
    assert tuple(got.shape) == (2, 8, 16, 16) and torch.equal(got[:, :4], vq0) and torch.equal(ref[:, :4], vq0)
    assert all(s_["c"].shape[1] >= 512 and torch.equal(s_["c"][1, :512], torch.arange(512)) for s_ in steps)    # fixed context: frames 0-1 of the window
    assert steps[0]["t"].shape[1] == 512                       # edit mode: only the last two latent frames of a window are ever targets
    same = bool(torch.equal(got, ref))
    why = None if same else _explain_difference(m, steps, hsteps, mask_rec, stream, 0.9, top_k=64)
    print(f"[c2 extrapolate edit=True, 2 jumps] {len(steps)} forwards; code map == oracle: {same} (proven ties: {why})")


def test_c5_pipeline_vs_the_two_oracles():
    """BASELINE config 5 composed at FULL geometry in fp32: a [1, 3, 16, 128, 128] pixel clip -> 3D-VQGAN encode ->
    token grid [1, 4, 16, 16] -> 8-step MaskGIT `sample` (cosine schedule, as the sampling script sets it) -> 1 x 2 `revise` passes
    (draft_and_revise with skip_draft) -> VQGAN decode, on the HIP path, against oracle/vqgan_oracle.py and oracle/mebt_oracle.py
    driven by the same permutations and Exp(1) noise.  Token ids bit-exact at every stage (a difference must be a provable fp tie
    of the oracle: its two best candidates within 5e-4); the decoded clip within 2e-5 of the oracle's decode of the same ids."""
    import argparse
    from mebt_amd.vqgan import VQGAN
    from oracle import vqgan_oracle as vq
    from tests.golden import make_golden as mg
    # ---- first stage (closed-form weights of the golden fixtures), transformer = the Sky / Taichi network, perturbed random init
    c = mg.VQGAN_CONFIGS["vq_c5"]
    args = argparse.Namespace(n_hiddens=c["n_hiddens"], downsample=c["downsample"], image_channels=3, embedding_dim=c["embedding_dim"],
                              n_codes=c["n_codes"], sequence_length=16, sample_every_n_frames=1, resolution=128)
    vqm = VQGAN(args)
    vcfg = mg.vqgan_cfg("vq_c5")
    VP = vq.closed_form_params(vcfg)
    vqm.load_state_dict(VP, strict=False)
    vqm.compute_dtype = "f32"
    vqm = vqm.to(DEV).eval()
    cfg = presets.taichi_16f()
    ocfg = oracle_cfg_of(cfg)
    sd = perturbed_state(41, cfg)
    m = presets.build_model(cfg, compute_dtype="f32")
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.mask_sampler.schedule = "cosine"                       # sample_vqgan_transformer_videos.py:189,219
    video = mg.vqgan_video("vq_c5")

    def ties_only(got, ref, what, allowed):
        n = int((got != ref).sum())
        assert n <= allowed, (what, n)
        return n

    # ---- 1. encode
    _, vq_ids = vqm.encode(video.to(DEV), include_embeddings=True)
    with torch.no_grad():
        ids_ref = vq.encode(VP, vcfg, video)
    assert tuple(vq_ids.shape) == (1, 4, 16, 16)
    ties_only(vq_ids.cpu(), ids_ref, "vqgan ids", 2)             # the reference's own fp32 near-ties (tests/test_gpu_vqgan.py)
    x_enc = ids_ref.clone()

    def stream(k, kind, shape):
        g = torch.Generator().manual_seed(515100 + k)
        if kind == "perm":
            return torch.randperm(int(shape[0]), generator=g)
        return torch.empty(tuple(shape), dtype=torch.float32).exponential_(generator=g)

    # ---- 2. sample: 8 steps from an empty context
    ctr = {"k": 0}

    def hook(kind, shape):
        k = ctr["k"]
        ctr["k"] += 1
        return stream(k, kind, shape)

    m.noise_hook = hook
    x0 = torch.zeros(1, 4, 16, 16, dtype=torch.long)
    got_s, ci_s, ti_s = m.sample(x0.to(DEV), None, 1.0, None, None, 8, None, None, context_temperature=2.0, skips=False)
    n_sample_draws = ctr["k"]
    octr = {"k": 0}

    def noise_fn(tag, shape):
        k = octr["k"]
        octr["k"] += 1
        return stream(k, "exp", shape)

    def perm_fn(tag, B_, N_):
        out = []
        for _ in range(B_):
            out.append(stream(octr["k"], "perm", (N_,)))
            octr["k"] += 1
        return torch.stack(out)

    steps, trace, mask_noise = [], [], []

    def rec_noise_fn(tag, shape):
        k = octr["k"]
        if tag == "sample":
            steps[-1]["noise_k"] = k
        nz = noise_fn(tag, shape)
        if tag == "mask":
            mask_noise.append(nz)
        return nz

    def logits_fn(partial, c_, t_):
        with torch.no_grad():
            lg = orc.reconstruct_mask(sd, ocfg, partial, c_, t_)
        steps.append({"partial": partial.clone(), "c": c_.clone(), "t": t_.clone(), "logits": lg})
        return lg

    with torch.no_grad():
        ref_s, rci, rti = orc.sample(sd, ocfg, x0, 8, 1.0, None, None, 2.0, rec_noise_fn, schedule="cosine", logits_fn=logits_fn, trace=trace)
    assert octr["k"] == n_sample_draws and len(steps) == len(trace) == len(mask_noise) == 8 and steps[0]["c"].shape[1] == 0
    n_tie = 0
    if torch.equal(got_s.cpu(), ref_s):
        assert torch.equal(ci_s.cpu(), rci) and torch.equal(ti_s.cpu(), rti)
    else:
        # The runs parted somewhere.  Every step is re-run on the HIP path FROM THE ORACLE'S STATE: the logits agree, sampled ids
        # differ only at provable fp ties, and the next-mask kernel's order is a descending order of the oracle's own keys
        # (score / sum) / q^ctemp up to 1e-5 relative — the ways two correct fp32 implementations may part on 1024 x 16384
        # candidates (here: two remaining targets whose keys tie swap rows and then receive each other's noise).
        from mebt_amd.transformer import sample_from_logits_scored
        for i, (st_, tr) in enumerate(zip(steps, trace)):
            lg, _ = m.reconstruct_mask(st_["partial"].to(DEV), st_["c"].to(DEV), st_["t"].to(DEV))
            assert (lg.cpu() - st_["logits"]).abs().max().item() < 1e-3
            nz = stream(st_["noise_k"], "exp", tuple(st_["logits"].shape))
            ids, sc, _ = sample_from_logits_scored(lg, 1.0, None, None, nz.to(DEV))
            for b, j in (ids.cpu() != tr["ids"]).nonzero().tolist():
                pr = torch.softmax(st_["logits"][b, j].double(), -1)
                top2 = (pr / nz[b, j].double()).topk(2).values
                assert top2[0] / top2[1] < 1 + 5e-4, ("not a tie", i, b, j)
                n_tie += 1
            assert ((sc.cpu() - tr["scores"]).abs() <= 2e-5 * tr["scores"])[ids.cpu() == tr["ids"]].all()
            NC_, NT_ = tr["NC"], tr["NT"]
            n_new = (NC_ + NT_ - tr["n_masked"]) - NC_
            if n_new <= 0:
                continue
            m.mask_sampler.noise_hook = lambda kind, shape, _nz=mask_noise[i]: _nz       # the oracle's draw of this step
            pc, pt = m.mask_sampler.generate_next_mask(tr["ci"].to(DEV), tr["ti"].to(DEV), tr["scores"].to(DEV), 0.0, strategy="maskgit",
                                                       context_temperature=tr["ctemp"], n_masked_toks=torch.tensor([float(tr["n_masked"])]))
            key = (tr["scores"] / tr["scores"].sum(-1, keepdim=True)).double() / mask_noise[i].double() ** tr["ctemp"]
            pos = {int(v): k_ for k_, v in enumerate(tr["ti"][0].tolist())}
            order = torch.cat([pc.cpu()[0, NC_:], pt.cpu()[0]])
            assert sorted(order.tolist()) == sorted(tr["ti"][0].tolist()) and torch.equal(pc.cpu()[0, :NC_], tr["ci"][0])
            ko = torch.stack([key[0, pos[int(v)]] for v in order])
            assert bool((ko[1:] <= ko[:-1] * (1 + 1e-5)).all()), ("next-mask order is not the oracle's up to fp ties", i)
        assert n_tie <= 3
    # ---- 3. revise: M = 1 pass of 2 steps over the oracle's sample (both sides start from the same tokens)
    start = ref_s.view(1, 4, 16, 16)
    ctr["k"] = octr["k"] = 1000
    steps.clear()
    got_r = m.draft_and_revise(start.to(DEV), None, 8, 1.0, None, None, 2, 0.3, None, None, 1, True).cpu()
    with torch.no_grad():
        ref_r = orc.draft_and_revise(sd, ocfg, start, 8, 1.0, None, None, 2, 0.3, None, None, 1, True, perm_fn, rec_noise_fn, logits_fn=logits_fn)
    assert ctr["k"] == octr["k"] and len(steps) == 2
    if not torch.equal(got_r, ref_r):
        n_tie += _replay_steps_prove_ties(m, steps, stream, 0.3)
    # ---- 4. decode the oracle's final tokens on both sides
    code = ref_r.view(1, 4, 16, 16)
    rec = vqm.decode(code.to(DEV)).cpu()
    with torch.no_grad():
        rec_ref = vq.decode(VP, vcfg, code)
    assert tuple(rec.shape) == tuple(video.shape)
    err = (rec - rec_ref).abs().max().item() / rec_ref.abs().max().item()
    print(f"[c5 pipeline f32] VQGAN ids differing {int((vq_ids.cpu() != ids_ref).sum())}; sampled == oracle: {bool(torch.equal(got_s.cpu(), ref_s))}, revised == oracle: "
          f"{bool(torch.equal(got_r, ref_r))} (proven fp ties: {n_tie}); decode rel err {err:.2e}")
    assert err < 2e-5
    assert int((x_enc != ref_r.view_as(x_enc)).sum()) > 0      # the pipeline did change tokens
