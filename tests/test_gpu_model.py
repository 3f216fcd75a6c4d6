"""Model-level parity of the HIP engine (C ABI: mebt_forward / mebt_loss / mebt_backward_* /
mebt_adamw_step) against the oracle and the committed golden vectors.  GPU only.

Tolerances: fp32 mode (exact-fp32 MFMA) is the north-star parity gate, logits within 1e-3 of the
reference CPU path (we measure ~1e-5); bf16 mode (the benchmarked precision) is checked at a
looser, stated tolerance."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mebt_oracle as orc
from tests.golden import make_golden as mg
from tests.helpers import build_native, load_golden, params_for

DEV = "cuda"
FP32_TOL = 1e-3          # BASELINE.json north_star: "within 1e-3 fp32"
FP32_TIGHT = 1e-4        # what the fp32-MFMA path actually achieves on these sizes
BF16_TOL = 1.2e-2        # bf16 compute vs fp32 oracle: 2x the 6.0e-3 measured on MI355X (|logits| ~ 0.3-1; torch autocast: 6.6e-3)
BF16_GRAD_TOL = 8e-2     # bf16 gradients, max error relative to the tensor's max |g| (measured 4.6e-2 at c1)


@pytest.mark.parametrize("name", ["c1", "micro", "micro_budget", "micro_maskgit"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_forward_logits_and_loss_vs_golden(name, dtype):
    g = load_golden("forward_" + name)
    cfg = mg.oracle_cfg(name)
    nm = build_native(cfg, dtype)
    x, idx = torch.from_numpy(g["x"]), torch.from_numpy(g["indices"])
    B = x.shape[0]
    tol = FP32_TIGHT if dtype == "f32" else BF16_TOL
    for c, (mode, t) in enumerate(zip(g["case_mode"], g["case_t"])):
        training = mode == "train"
        ci, ti, seq_len = orc.divide_indices(idx, float(t), cfg, training)
        logits = nm.forward(x.reshape(B, -1).to(DEV), ci.to(DEV), ti.to(DEV), training=True)   # engine keeps refs
        st = nm.loss_stats(logits).cpu()
        lg = logits.cpu()
        e_cols = np.abs(lg[..., g["cols"]].numpy() - g[f"c{c}_cols"]).max()
        print(f"[fwd {name} {dtype} case {c}] max |dlogits| {e_cols:.3e}")
        np.testing.assert_allclose(lg[..., g["cols"]].numpy(), g[f"c{c}_cols"], atol=tol, rtol=0)
        np.testing.assert_allclose(torch.logsumexp(lg.double(), -1).numpy(), g[f"c{c}_lse"], atol=tol, rtol=0)
        meta = g[f"c{c}_meta"]
        ntw = meta[0]
        loss = float(st[0]) / (B * seq_len * (ntw / seq_len) ** cfg.avg_loss)
        assert abs(loss - meta[2]) < (2e-5 if dtype == "f32" else 2e-3) * abs(meta[2]), (loss, meta[2])
        if dtype == "f32":
            n = int(st[3])
            assert abs(100.0 * float(st[1]) / n - meta[3]) < 1e-3 and abs(100.0 * float(st[2]) / n - meta[4]) < 1e-3
            assert (lg.argmax(-1).numpy() == g[f"c{c}_argmax"]).mean() > 0.99


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_edges_nc0_nt1(dtype):
    e = load_golden("edges_micro")
    cfg = mg.oracle_cfg("micro")
    nm = build_native(cfg, dtype)
    x, idx = torch.from_numpy(e["x"]).reshape(2, -1).to(DEV), torch.from_numpy(e["indices"]).to(DEV)
    tol = FP32_TIGHT if dtype == "f32" else BF16_TOL
    l0 = nm.forward(x, idx[:, :0].contiguous(), idx, training=False).cpu()          # NC = 0
    assert torch.isfinite(l0).all()
    np.testing.assert_allclose(l0[..., e["cols"]].numpy(), e["nc0_cols"], atol=tol, rtol=0)
    l1 = nm.forward(x, idx[:, :-1].contiguous(), idx[:, -1:].contiguous(), training=False).cpu()   # NT = 1
    np.testing.assert_allclose(l1[..., e["cols"]].numpy(), e["nt1_cols"], atol=tol, rtol=0)


@pytest.mark.parametrize("name,dtype,side", [("micro", "f32", 0), ("c1", "f32", 0), ("micro", "bf16", 0), ("c1", "bf16", 0),
                                             ("c1", "f32", 1), ("c1", "bf16", 1)])
def test_gradients_vs_oracle_autograd(name, dtype, side):
    """side = 1: gradient leaves on the second stream (MEBT_SIDE_STREAM=1 path)"""
    cfg = mg.oracle_cfg(name)
    P = {k: v.clone().requires_grad_(True) for k, v in params_for(name).items()}
    nm = build_native(cfg, dtype)
    nm.lib.mebt_debug_side_stream(nm.h, side)
    B = 3
    x, idx = mg.inputs(name, B, "grad")
    for t in (0.5, 0.2, 0.0):      # t = 0 -> NC = 0
        for p in P.values():
            p.grad = None
        logits, z_t, ntw, seq_len = orc.forward(P, cfg, x, idx, t, training=True)
        _, _, loss = orc.loss_and_acc(logits, z_t, ntw, seq_len, cfg)
        loss.backward()
        ci, ti, _ = orc.divide_indices(idx, t, cfg, True)
        lg = nm.forward(x.reshape(B, -1).to(DEV), ci.to(DEV), ti.to(DEV), training=True)
        scale = 1.0 / (B * seq_len * (ntw / seq_len) ** cfg.avg_loss)
        nm.backward(lg, scale)
        torch.cuda.synchronize()
        gv = nm.views(orc.param_shapes(cfg), grads=True)
        lim = 2e-3 if dtype == "f32" else BF16_GRAD_TOL
        bad, worst = [], 0.0
        for k, p in P.items():
            ref = p.grad if p.grad is not None else torch.zeros_like(p)
            got = gv[k].cpu()
            denom = ref.abs().max().item() + 1e-6      # attn.key.bias has an exactly-zero gradient (softmax shift invariance)
            err = (got - ref).abs().max().item() / denom
            worst = max(worst, err)
            if not err < lim:
                bad.append((k, round(err, 5), denom))
        print(f"[grad {name} {dtype} side {side} t {t}] worst rel-to-max {worst:.3e}")
        assert not bad, (name, dtype, t, len(bad), bad[:40])


@pytest.mark.parametrize("name", ["micro", "micro_budget"])
def test_train_steps_vs_golden(name):
    """forward + loss + backward + fused AdamW for three steps == the reference's Lightning step
    (golden: losses, per-parameter norms after each step)."""
    g = load_golden("train_" + name)
    cfg = mg.oracle_cfg(name)
    nm = build_native(cfg, "f32")
    names = [str(n) for n in g["names"]]
    for s, t in enumerate(g["ts"]):
        x, idx = torch.from_numpy(g[f"s{s}_x"]), torch.from_numpy(g[f"s{s}_indices"])
        B = x.shape[0]
        ci, ti, seq_len = orc.divide_indices(idx, float(t), cfg, True)
        ntw = float(seq_len - ci.shape[1])
        lg = nm.forward(x.reshape(B, -1).to(DEV), ci.to(DEV), ti.to(DEV), training=True)
        st = nm.loss_stats(lg)
        scale = 1.0 / (B * seq_len * (ntw / seq_len) ** cfg.avg_loss)
        nm.backward(lg, scale)
        loss = float(st[0].cpu()) * scale
        meta = g[f"s{s}_meta"]
        assert abs(loss - meta[0]) < 5e-5 * abs(meta[0]), (s, loss, meta[0])
        gv = nm.views(orc.param_shapes(cfg), grads=True)
        gn = np.array([float(gv[n].double().norm().cpu()) for n in names])
        np.testing.assert_allclose(gn, g[f"s{s}_gradnorm"], rtol=2e-3, atol=1e-6)
        nm.adamw_step(float(g["lr"]), float(g["wd"]), s + 1)
        pv = nm.views(orc.param_shapes(cfg))
        pn = np.array([float(pv[n].double().norm().cpu()) for n in names])
        np.testing.assert_allclose(pn, g[f"s{s}_pnorm"], rtol=2e-5)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_maskgit_blocks_rewrite_both_streams(dtype, drop):
    """'maskgit' blocks (the padding mode of reference gpt.py:176-178,191-192,208-209) in the MIDDLE of a network:
    they rewrite contexts and targets, so later latent_enc blocks read the rewritten contexts and the gradient of the
    contexts stream is accumulated / handed back across them.  Forward, eval forward and every gradient vs the oracle
    (with the kernels' own dropout masks injected when drop > 0)."""
    from mebt_amd import _lib
    from mebt_amd.engine import NativeModel
    modes = ["latent_enc", "maskgit", "latent_self", "latent_enc", "latent_dec", "lt2l", "maskgit", "latent_enc", "latent_dec", "maskgit"]
    cfg = orc.OracleConfig(len(modes), 2, 64, 32, 8, modes, shape=[2, 4, 4], budget=32, avg_loss=1.0)
    nm = NativeModel(cfg.n_layer, cfg.n_head, cfg.n_embd, cfg.vocab_size, cfg.sos_emb, cfg.block_size, cfg.mode, dtype=dtype,
                     embd_pdrop=drop, resid_pdrop=drop, attn_pdrop=drop)
    nm.allocate(DEV)
    P0 = orc.closed_form_params(cfg)
    views = nm.views(orc.param_shapes(cfg))
    with torch.no_grad():
        for k, v in P0.items():
            views[k].copy_(v)
    nm.sync_lowp(force=True)
    site = {"attn": 0, "proj": 1, "mlp": 2, "emb_sos": 0xFFFF0, "emb_ctx": 0xFFFF1, "emb_tgt": 0xFFFF2}
    seed = 0x5EED1234

    def kmask(kind, layer, tensor):
        if drop == 0.0 or tensor.numel() == 0:
            return tensor
        sid = site[kind] if kind.startswith("emb") else 16 * layer + site[kind]
        n = tensor.numel()
        pad = (-n) % 4
        out = torch.ones(n + pad, device=DEV)
        _lib.check(_lib.load().mebt_debug_dropout_mask(seed, sid, drop, n + pad, _lib.ptr(out), _lib.cur_stream()))
        torch.cuda.synchronize()
        return tensor * out[:n].cpu().view(tensor.shape)

    B = 3
    x, idx = mg.inputs("micro", B, "maskgit-mid")
    for t in (0.5, 0.15, 0.0):                   # t = 0 -> NC = 0: the maskgit blocks attend over the targets only
        P = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
        logits, z_t, ntw, seq_len = orc.forward(P, cfg, x, idx, t, training=True, drop=kmask)
        _, _, loss = orc.loss_and_acc(logits, z_t, ntw, seq_len, cfg)
        loss.backward()
        ci, ti, _ = orc.divide_indices(idx, t, cfg, True)
        lg = nm.forward(x.reshape(B, -1).to(DEV), ci.to(DEV), ti.to(DEV), training=True, dropout_seed=seed)
        tol = FP32_TIGHT if dtype == "f32" else 3 * BF16_TOL        # 10 blocks of d = 64: |logits| ~ 3x the 4-6 block configs
        assert (lg.cpu() - logits.detach()).abs().max().item() < tol, t
        scale = 1.0 / (B * seq_len * (ntw / seq_len))
        nm.backward(lg, scale)
        torch.cuda.synchronize()
        gv = nm.views(orc.param_shapes(cfg), grads=True)
        lim = 2e-3 if dtype == "f32" else 1.2e-1
        bad = []
        for k, p in P.items():
            ref = p.grad if p.grad is not None else torch.zeros_like(p)
            err = (gv[k].cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-5)
            if not err < lim:
                bad.append((k, round(err, 5)))
        assert not bad, (t, bad[:20])
        ev = nm.forward(x.reshape(B, -1).to(DEV), ci.to(DEV), ti.to(DEV), training=False)
        with torch.no_grad():
            ref_eval = orc.reconstruct_mask(P0, cfg, x.reshape(B, -1), ci, ti)
        assert (ev.cpu() - ref_eval).abs().max().item() < tol


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_all_maskgit_without_latents(dtype):
    """`sos_emb = 0` (transformer.py:273-276 gives a [B,0,d] latent tensor) with every block in the
    'maskgit' full-attention mode — a plain MaskGIT transformer: forward, eval forward and gradients vs the oracle; a latent
    routing mode together with sos_emb = 0 is refused."""
    from mebt_amd.engine import NativeModel
    from mebt_amd import _lib
    modes = ["maskgit"] * 3
    cfg = orc.OracleConfig(3, 2, 64, 32, 0, modes, shape=[2, 4, 4], budget=32, avg_loss=1.0)
    nm = NativeModel(cfg.n_layer, cfg.n_head, cfg.n_embd, cfg.vocab_size, 0, cfg.block_size, cfg.mode, dtype=dtype)
    nm.allocate(DEV)
    P0 = orc.closed_form_params(cfg)
    views = nm.views(orc.param_shapes(cfg))
    with torch.no_grad():
        for k, v in P0.items():
            views[k].copy_(v)
    nm.sync_lowp(force=True)
    x, idx = mg.inputs("micro", 2, "nosos")
    for t in (0.4, 0.0):
        P = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
        logits, z_t, ntw, seq_len = orc.forward(P, cfg, x, idx, t, training=True)
        _, _, loss = orc.loss_and_acc(logits, z_t, ntw, seq_len, cfg)
        loss.backward()
        ci, ti, _ = orc.divide_indices(idx, t, cfg, True)
        lg = nm.forward(x.reshape(2, -1).to(DEV), ci.to(DEV), ti.to(DEV), training=True)
        tol = FP32_TIGHT if dtype == "f32" else 2 * BF16_TOL
        assert (lg.cpu() - logits.detach()).abs().max().item() < tol
        nm.backward(lg, 1.0 / (2 * seq_len * (ntw / seq_len)))
        torch.cuda.synchronize()
        gv = nm.views(orc.param_shapes(cfg), grads=True)
        lim = 2e-3 if dtype == "f32" else BF16_GRAD_TOL
        for k, p in P.items():
            ref = p.grad if p.grad is not None else torch.zeros_like(p)
            scale = P[k.replace("attn.key.bias", "attn.query.bias")].grad.abs().max().item() + 1e-9
            assert (gv[k].cpu() - ref).abs().max().item() / scale < lim, (t, k)
    with pytest.raises(_lib.MebtError, match="maskgit"):
        NativeModel(2, 2, 64, 16384, 0, 32, ["latent_enc", "maskgit"], dtype=dtype)
