"""Dropout (embd / attn / resid, reference gpt.py:135,140,154,238-240) in the HIP engine.
The masks are counter-based and recomputed in backward; the test reads the exact masks the kernels
use (mebt_debug_dropout_mask), injects them into the oracle at the reference's dropout sites and
compares logits and every parameter gradient.  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mebt_amd import _lib
from mebt_amd.engine import NativeModel
from oracle import mebt_oracle as orc
from tests.golden import make_golden as mg

DEV = "cuda"
SITE = {"attn": 0, "proj": 1, "mlp": 2, "emb_sos": 0xFFFF0, "emb_ctx": 0xFFFF1, "emb_tgt": 0xFFFF2}


def kernel_mask(seed, site, p, shape):
    n = int(np.prod(shape))
    pad = (-n) % 4
    out = torch.ones(n + pad, device=DEV)
    _lib.check(_lib.load().mebt_debug_dropout_mask(seed, site, p, n + pad, _lib.ptr(out), _lib.cur_stream()))
    torch.cuda.synchronize()
    return out[:n].cpu().view(*shape)


@pytest.mark.parametrize("name,dtype", [("micro", "f32"), ("c1", "f32"), ("c1", "bf16")])
def test_dropout_sites_forward_backward(name, dtype):
    cfg = mg.oracle_cfg(name)
    p_emb, p_res, p_att = 0.1, 0.2, 0.15
    nm = NativeModel(cfg.n_layer, cfg.n_head, cfg.n_embd, cfg.vocab_size, cfg.sos_emb, cfg.block_size, cfg.mode,
                     dtype=dtype, embd_pdrop=p_emb, resid_pdrop=p_res, attn_pdrop=p_att)
    nm.allocate(DEV)
    P = orc.closed_form_params(cfg)
    views = nm.views(orc.param_shapes(cfg))
    with torch.no_grad():
        for k, v in P.items():
            views[k].copy_(v)
    nm.sync_lowp(force=True)
    B, seed, t = 2, 0x1234ABCD5, 0.45
    x, idx = mg.inputs(name, B, "drop")
    ci, ti, seq_len = orc.divide_indices(idx, t, cfg, True)
    pk = {"attn": p_att, "proj": p_res, "mlp": p_res, "emb_sos": p_emb, "emb_ctx": p_emb, "emb_tgt": p_emb}
    seen = []

    def drop(kind, layer, tensor):
        site = SITE[kind] if kind.startswith("emb") else 16 * layer + SITE[kind]
        m = kernel_mask(seed, site, pk[kind], tuple(tensor.shape))
        seen.append((kind, layer, float((m > 0).float().mean())))
        return tensor * m

    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    logits, z_t, ntw, _ = orc.forward(Pg, cfg, x, idx, t, training=True, drop=drop)
    _, _, loss = orc.loss_and_acc(logits, z_t, ntw, seq_len, cfg)
    loss.backward()
    # keep rates are what p says
    for kind, layer, keep in seen:
        assert abs(keep - (1 - pk[kind])) < 0.06, (kind, layer, keep)

    lg = nm.forward(x.reshape(B, -1).to(DEV), ci.to(DEV), ti.to(DEV), training=True, dropout_seed=seed)
    tol = 1e-4 if dtype == "f32" else 2e-2
    assert (lg.cpu() - logits.detach()).abs().max().item() < tol
    scale = 1.0 / (B * seq_len * (ntw / seq_len))
    nm.backward(lg, scale)
    torch.cuda.synchronize()
    gv = nm.views(orc.param_shapes(cfg), grads=True)
    lim = 2e-3 if dtype == "f32" else 1e-1
    bad = []
    for k, pr in Pg.items():
        ref = pr.grad if pr.grad is not None else torch.zeros_like(pr)
        err = (gv[k].cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-5)   # key.bias: exactly-zero gradient
        if not err < lim:
            bad.append((k, round(err, 5)))
    assert not bad, bad[:20]
    # a different seed gives a different network function; eval mode ignores dropout entirely
    lg2 = nm.forward(x.reshape(B, -1).to(DEV), ci.to(DEV), ti.to(DEV), training=True, dropout_seed=seed + 1)
    assert (lg2 - lg).abs().max().item() > 1e-3
    lg3 = nm.forward(x.reshape(B, -1).to(DEV), ci.to(DEV), ti.to(DEV), training=False)
    ref_eval = orc.reconstruct_mask(P, cfg, x, ci, ti)
    assert (lg3.cpu() - ref_eval).abs().max().item() < tol


def test_fused_doutm_is_bit_identical():
    """MEBT_FUSE_DOUTM (the LN1-backward launch of block i also writes block i-1's dropout-masked output gradient, bf16
    only, on by default) against the separate elementwise launch: every gradient identical except the atomically
    accumulated P-side ones, which agree to fp32 summation order.  Runs the two settings in child processes (the switch
    is read once per process)."""
    import os, subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
from mebt_amd.engine import NativeModel
from oracle import mebt_oracle as orc
from tests.golden import make_golden as mg
cfg = mg.oracle_cfg("c1")
nm = NativeModel(cfg.n_layer, cfg.n_head, cfg.n_embd, cfg.vocab_size, cfg.sos_emb, cfg.block_size, cfg.mode, dtype="bf16",
                 embd_pdrop=0.1, resid_pdrop=0.2, attn_pdrop=0.15)
nm.allocate("cuda")
views = nm.views(orc.param_shapes(cfg))
with torch.no_grad():
    for k, v in orc.closed_form_params(cfg).items():
        views[k].copy_(v)
nm.sync_lowp(force=True)
x, idx = mg.inputs("c1", 2, "drop")
ci, ti, seq_len = orc.divide_indices(idx, 0.45, cfg, True)
lg = nm.forward(x.reshape(2, -1).cuda(), ci.cuda(), ti.cuda(), training=True, dropout_seed=77)
nm.backward(lg, 1.0 / (2 * seq_len))
torch.cuda.synchronize()
np.savez(sys.argv[1], gW=nm.gW.cpu().numpy(), gP=nm.gP.cpu().numpy())
''' % root
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for flag in ("0", "1"):
            path = os.path.join(td, f"g{flag}.npz")
            env = dict(os.environ, MEBT_FUSE_DOUTM=flag, MEBT_GEMM_AUTOTUNE="0")
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, cwd=root)
            outs.append(np.load(path))
    assert np.array_equal(outs[0]["gW"], outs[1]["gW"])
    scale = np.abs(outs[0]["gP"]).max()
    assert np.abs(outs[0]["gP"] - outs[1]["gP"]).max() <= 1e-5 * scale


def test_dropout_mask_statistics():
    """The counter-based masks (one 32-bit hash per PAIR of elements, 16 bits each, csrc/common.h) as a random source:
    keep rate within 4 sigma of 1 - p, and no linear dependence between (i) the two halves of a hash (elements 2k, 2k+1),
    (ii) neighbours at lags 1, 2, 64, 1024, (iii) the same element at different sites / layers, (iv) consecutive seeds —
    every correlation within 5 / sqrt(N) of zero (N = 4 M elements)."""
    n = 1 << 22
    p = 0.1

    def keep(seed, site):
        return (kernel_mask(seed, site, p, (n,)) > 0).float()

    base = keep(0xABCDEF0123, 16 * 3 + SITE["mlp"])
    sd = np.sqrt(p * (1 - p) / n)
    assert abs(base.mean().item() - (1 - p)) < 4 * sd
    lim = 5.0 / np.sqrt(n)

    def corr(a, b):
        a, b = a - a.mean(), b - b.mean()
        return float((a * b).mean() / (a.std() * b.std() + 1e-12))

    assert abs(corr(base[0::2], base[1::2])) < lim * np.sqrt(2)               # the two 16-bit halves of one hash
    for lag in (1, 2, 64, 1024):
        assert abs(corr(base[:-lag], base[lag:])) < lim, lag
    for other in (16 * 3 + SITE["proj"], 16 * 3 + SITE["attn"], 16 * 4 + SITE["mlp"], SITE["emb_tgt"]):
        assert abs(corr(base, keep(0xABCDEF0123, other))) < lim, other
    assert abs(corr(base, keep(0xABCDEF0124, 16 * 3 + SITE["mlp"]))) < lim
    assert abs(corr(base, keep(0xABCDEF0123 + (1 << 20), 16 * 3 + SITE["mlp"]))) < lim     # the next global step's seed (transformer.py _next_seed)
