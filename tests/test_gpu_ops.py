"""Operator-level parity of the HIP kernels (through the C ABI) against fp32/fp64 CPU references.
GPU only.  fp32 mode must agree to ~1e-5 (exact-fp32 MFMA), bf16 mode to bf16 rounding."""
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream

DEV = "cuda"


def lib():
    return _lib.load()


def tdt(dtype):
    return torch.bfloat16 if dtype == _lib.BF16 else torch.float32


def rnd(*shape, seed=0, scale=1.0, ints=False):
    g = torch.Generator().manual_seed(seed)
    if ints:   # small integers: exact in bf16 and in fp32 accumulation -> layout bugs show as exact mismatches
        return torch.randint(-3, 4, shape, generator=g).float()
    return torch.randn(*shape, generator=g) * scale


def run_gemm(dtype, A, B, M, N, K, a_kc, b_kc, bias=None, aux=None, epilogue=0, c_f32=0, beta=0, split_k=1, C0=None):
    """A, B: CPU fp32 tensors already in their storage layout."""
    t = tdt(dtype)
    Ad, Bd = A.to(DEV, t).contiguous(), B.to(DEV, t).contiguous()
    out_t = torch.float32 if (c_f32 or dtype == _lib.F32) else t
    Cd = torch.full((M, N), float("nan"), device=DEV, dtype=out_t) if C0 is None else C0.to(DEV, out_t).clone()
    C2d = torch.full((M, N), float("nan"), device=DEV, dtype=t) if epilogue == _lib.EPI_GELU else None
    bd = bias.to(DEV) if bias is not None else None
    xd = aux.to(DEV, t).contiguous() if aux is not None else None
    lda = A.shape[1]
    ldb = B.shape[1]
    check(lib().mebt_op_gemm(dtype, ptr(Ad), ptr(Bd), ptr(Cd), ptr(C2d), ptr(bd), ptr(xd), M, N, K, lda, ldb, N, N,
                             int(a_kc), int(b_kc), epilogue, c_f32, beta, split_k, cur_stream()))
    torch.cuda.synchronize()
    return Cd.float().cpu(), (C2d.float().cpu() if C2d is not None else None)


def q(x, dtype):
    """round operands like the device path does"""
    return x.to(tdt(dtype)).float()


@pytest.mark.parametrize("dtype", [_lib.BF16, _lib.F32])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 0), (0, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (200, 136, 128), (72, 64, 320)])
def test_gemm_layouts_exact_integers(dtype, a_kc, b_kc, M, N, K):
    """Asymmetric small-integer operands: every layout / fragment / swizzle mistake is an exact mismatch."""
    if not a_kc and M % 8:
        pytest.skip("RC operand rows must be a multiple of 8")
    Am, Bm = rnd(M, K, seed=1, ints=True), rnd(N, K, seed=2, ints=True)
    ref = Am.double() @ Bm.double().t()
    A = Am if a_kc else Am.t().contiguous()
    B = Bm if b_kc else Bm.t().contiguous()
    out, _ = run_gemm(dtype, A, B, M, N, K, a_kc, b_kc, c_f32=1)
    assert torch.equal(out.double(), ref), (out.double() - ref).abs().max()


@pytest.mark.parametrize("staging", [0, 2, 1, 10, 11, 12])
@pytest.mark.parametrize("tile", [(128, 128), (128, 64), (64, 128), (64, 64), (192, 128), (96, 128), (96, 64), (256, 256)])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 0), (0, 1)])
def test_gemm_every_tile_variant(tile, a_kc, b_kc, staging):
    """each block-tile x staging instantiation of the bf16 kernel (register-staged, LDS-DMA 2-stage,
    LDS-DMA 3-stage ring; 8 + ring: the software-pipelined main loop, rings 2-4), forced, on ragged integer operands"""
    M, N, K = 328, 200, (320 if staging >= 8 else 192)
    if staging >= 8 and tile == (256, 256):
        pytest.skip("the 8-wave tile has no pipelined variant")
    Am, Bm = rnd(M, K, seed=11, ints=True), rnd(N, K, seed=12, ints=True)
    ref = Am.double() @ Bm.double().t()
    A = Am if a_kc else Am.t().contiguous()
    B = Bm if b_kc else Bm.t().contiguous()
    lib().mebt_debug_gemm_tile(*tile)
    lib().mebt_debug_gemm_variant(staging)
    try:
        out, _ = run_gemm(_lib.BF16, A, B, M, N, K, a_kc, b_kc, c_f32=1)
    finally:
        lib().mebt_debug_gemm_tile(0, 0)
        lib().mebt_debug_gemm_variant(-1)
    assert torch.equal(out.double(), ref), (out.double() - ref).abs().max()


@pytest.mark.parametrize("b_kc", [1, 0])
@pytest.mark.parametrize("M,N,K", [(328, 200, 64), (256, 256, 128), (520, 776, 192), (300, 264, 320), (1024, 512, 1024), (513, 1032, 2048)])
def test_gemm_staggered_groups_256(M, N, K, b_kc):
    """the 256 x 256 tile as two staggered 4-wave groups (variant 9; A row-major-in-k only): every k-tile count from the
    prologue-only cases (1, 2) through the refill schedule's steady state, ragged M / N, exact on integers"""
    Am, Bm = rnd(M, K, seed=31, ints=True), rnd(N, K, seed=32, ints=True)
    ref = Am.double() @ Bm.double().t()
    B = Bm if b_kc else Bm.t().contiguous()
    lib().mebt_debug_gemm_tile(256, 256)
    lib().mebt_debug_gemm_variant(9)
    try:
        out, _ = run_gemm(_lib.BF16, Am, B, M, N, K, 1, b_kc, c_f32=1)
    finally:
        lib().mebt_debug_gemm_tile(0, 0)
        lib().mebt_debug_gemm_variant(-1)
    assert torch.equal(out.double(), ref), (out.double() - ref).abs().max()


@pytest.mark.parametrize("b_kc", [1, 0])
@pytest.mark.parametrize("out", ["f32", "bf16", "bf16+bias+gelu", "bf16 ragged"])
def test_gemm_staggered_groups_256_persistent_tile_loop(out, b_kc):
    """more than 256 tiles: the workgroups of the staggered 256 x 256 kernel loop over tiles, issue the next tile's prologue in
    front of the epilogue and wait for it with a counted vmcnt (16 bf16 / 32 fp32 stores stay in flight) — the fp32 result is exact
    on integers, the bf16 one to its rounding; a GELU epilogue and a ragged edge take the uncounted path inside the same loop"""
    M, N, K = (4608, 4352, 192) if out != "bf16 ragged" else (4600, 4360, 192)          # 18 x 17 = 306 tiles (18 x 18 ragged)
    Am, Bm = rnd(M, K, seed=51, ints=True), rnd(N, K, seed=52, ints=True)
    ref = Am.double() @ Bm.double().t()
    B = Bm if b_kc else Bm.t().contiguous()
    bias = rnd(N, seed=53, ints=True) if "bias" in out else None
    lib().mebt_debug_gemm_tile(256, 256)
    lib().mebt_debug_gemm_variant(9)
    try:
        if out == "f32":
            got, _ = run_gemm(_lib.BF16, Am, B, M, N, K, 1, b_kc, c_f32=1)
            assert torch.equal(got.double(), ref), (got.double() - ref).abs().max()
        elif "gelu" in out:
            got, got2 = run_gemm(_lib.BF16, Am * 0.25, B, M, N, K, 1, b_kc, bias=bias, epilogue=_lib.EPI_GELU)
            pre = ref * 0.25 + bias.double()
            assert (got.double() - pre).abs().max() <= pre.abs().max() / 128
            g = F.gelu(pre.float()).double()
            assert (got2.double() - g).abs().max() <= g.abs().max() / 100
        else:
            got, _ = run_gemm(_lib.BF16, Am, B, M, N, K, 1, b_kc)
            assert (got.double() - ref).abs().max() <= ref.abs().max() / 256 and torch.equal(got.double()[ref.abs() <= 256], ref[ref.abs() <= 256])
    finally:
        lib().mebt_debug_gemm_tile(0, 0)
        lib().mebt_debug_gemm_variant(-1)


@pytest.mark.parametrize("ring", [2, 3])
@pytest.mark.parametrize("tile", [(96, 64), (64, 64), (96, 128), (64, 128), (128, 64)])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 0), (0, 1)])
def test_gemm_two_pipeline_variants(tile, a_kc, b_kc, ring):
    """8-wave workgroups running two 4-wave pipelines on alternate k-tiles (variant 16 + ring depth), ragged M / N"""
    M, N, K = 328, 200, 512
    Am, Bm = rnd(M, K, seed=21, ints=True), rnd(N, K, seed=22, ints=True)
    ref = Am.double() @ Bm.double().t()
    A = Am if a_kc else Am.t().contiguous()
    B = Bm if b_kc else Bm.t().contiguous()
    lib().mebt_debug_gemm_tile(*tile)
    lib().mebt_debug_gemm_variant(16 + ring)
    try:
        out, _ = run_gemm(_lib.BF16, A, B, M, N, K, a_kc, b_kc, c_f32=1)
    finally:
        lib().mebt_debug_gemm_tile(0, 0)
        lib().mebt_debug_gemm_variant(-1)
    assert torch.equal(out.double(), ref), (out.double() - ref).abs().max()


@pytest.mark.parametrize("dtype", [_lib.BF16, _lib.F32])
def test_gemm_ragged_reduction_and_splitk(dtype):
    """wgrad shape: reduction over an arbitrary token count (not a tile multiple), split-K atomics."""
    M, N, K = 256, 128, 333
    Am, Bm = rnd(M, K, seed=3, ints=True), rnd(N, K, seed=4, ints=True)
    ref = Am.double() @ Bm.double().t()
    for split in (1, 3, 0):
        out, _ = run_gemm(dtype, Am.t().contiguous(), Bm.t().contiguous(), M, N, K, 0, 0, c_f32=1, split_k=split)
        assert torch.equal(out.double(), ref), split
    # beta accumulate
    C0 = rnd(M, N, seed=5, ints=True)
    out, _ = run_gemm(dtype, Am.t().contiguous(), Bm.t().contiguous(), M, N, K, 0, 0, c_f32=1, beta=1, C0=C0)
    assert torch.equal(out.double(), ref + C0.double())


@pytest.mark.parametrize("dtype,tol", [(_lib.BF16, 2e-2), (_lib.F32, 2e-5)])
def test_gemm_epilogues(dtype, tol):
    M, N, K = 192, 256, 128
    A, B = q(rnd(M, K, seed=6), dtype), q(rnd(N, K, seed=7, scale=0.1), dtype)
    bias = rnd(N, seed=8)
    aux = q(rnd(M, N, seed=9), dtype)
    lin = A.double() @ B.double().t() + bias.double()
    out, _ = run_gemm(dtype, A, B, M, N, K, 1, 1, bias=bias)
    assert (out.double() - lin).abs().max() < tol * max(1, lin.abs().max())
    out, g = run_gemm(dtype, A, B, M, N, K, 1, 1, bias=bias, epilogue=_lib.EPI_GELU)
    assert (out.double() - lin).abs().max() < tol * max(1, lin.abs().max())
    assert (g.double() - F.gelu(lin)).abs().max() < tol * max(1, lin.abs().max())
    out, _ = run_gemm(dtype, A, B, M, N, K, 1, 1, bias=bias, aux=aux, epilogue=_lib.EPI_RESID)
    assert (out.double() - (lin + aux.double())).abs().max() < tol * max(1, lin.abs().max())
    x = aux.double().requires_grad_(True)
    F.gelu(x).sum().backward()
    nob = A.double() @ B.double().t()
    out, _ = run_gemm(dtype, A, B, M, N, K, 1, 1, aux=aux, epilogue=_lib.EPI_GELU_BWD)
    assert (out.double() - nob * x.grad).abs().max() < tol * max(1, nob.abs().max())


@pytest.mark.parametrize("variant", [34, 67])
@pytest.mark.parametrize("tile", [(128, 128), (192, 128)])
@pytest.mark.parametrize("b_kc", [1, 0])
def test_gemm_splitk_slabs_with_epilogues(tile, variant, b_kc):
    """split-K into fp32 slabs + reduce/epilogue kernel (variant 32 + ring: 2 ways, 64 + ring: 4 ways), forced"""
    M, N, K = 200, 264, 1024
    tol = 2e-2
    A = q(rnd(M, K, seed=6), _lib.BF16)
    Bm = q(rnd(N, K, seed=7, scale=0.1), _lib.BF16)
    B = Bm if b_kc else Bm.t().contiguous()
    bias = rnd(N, seed=8)
    aux = q(rnd(M, N, seed=9), _lib.BF16)
    lin = A.double() @ Bm.double().t() + bias.double()
    scratch = torch.empty(496 << 20, dtype=torch.uint8, device=DEV)      # caller-owned tuner / split-K scratch of the operator-level entry
    lib().mebt_debug_gemm_scratch(ptr(scratch), scratch.numel())
    lib().mebt_debug_gemm_tile(*tile)
    lib().mebt_debug_gemm_variant(variant)
    try:
        out, _ = run_gemm(_lib.BF16, A, B, M, N, K, 1, b_kc, bias=bias)
        assert (out.double() - lin).abs().max() < tol * max(1, lin.abs().max())
        assert scratch[384 << 20:(384 << 20) + 4 * M * N].view(torch.float32).abs().sum().item() > 0     # the slabs were really used
        out, g = run_gemm(_lib.BF16, A, B, M, N, K, 1, b_kc, bias=bias, epilogue=_lib.EPI_GELU)
        assert (out.double() - lin).abs().max() < tol * max(1, lin.abs().max())
        assert (g.double() - F.gelu(lin)).abs().max() < tol * max(1, lin.abs().max())
        out, _ = run_gemm(_lib.BF16, A, B, M, N, K, 1, b_kc, bias=bias, aux=aux, epilogue=_lib.EPI_RESID)
        assert (out.double() - (lin + aux.double())).abs().max() < tol * max(1, lin.abs().max())
    finally:
        lib().mebt_debug_gemm_tile(0, 0)
        lib().mebt_debug_gemm_variant(-1)
        torch.cuda.synchronize()
        lib().mebt_debug_gemm_scratch(None, 0)


@pytest.mark.parametrize("dtype,tol", [(_lib.BF16, 2e-2), (_lib.F32, 1e-5)])
@pytest.mark.parametrize("rows,d", [(37, 64), (130, 256), (64, 1024), (5, 320)])
def test_layernorm_fwd_bwd(dtype, tol, rows, d):
    t = tdt(dtype)
    x = q(rnd(rows, d, seed=1) * 2 + 0.3, dtype)
    w, b = 1 + 0.1 * rnd(d, seed=2), 0.1 * rnd(d, seed=3)
    dy = q(rnd(rows, d, seed=4), dtype)
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    y = F.layer_norm(xr, (d,), wr, br, 1e-5)
    (y * dy.double()).sum().backward()
    xd, yd = x.to(DEV, t), torch.empty(rows, d, device=DEV, dtype=t)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    wd, bd, dyd = w.to(DEV), b.to(DEV), dy.to(DEV, t)      # keep device tensors alive across the calls
    check(lib().mebt_op_layernorm_fwd(dtype, ptr(xd), ptr(yd), ptr(wd), ptr(bd), ptr(mean), ptr(rstd), rows, d, cur_stream()))
    assert (yd.float().cpu().double() - y.detach()).abs().max() < tol * 4
    dx = torch.empty(rows, d, device=DEV, dtype=t)
    dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    check(lib().mebt_op_layernorm_bwd(dtype, ptr(xd), ptr(dyd), ptr(wd), ptr(mean), ptr(rstd), ptr(dx), ptr(dg), ptr(db), rows, d, cur_stream()))
    torch.cuda.synchronize()
    assert (dx.float().cpu().double() - xr.grad).abs().max() < tol * 8
    assert (dg.cpu().double() - wr.grad).abs().max() < tol * rows ** 0.5 * 4 + 1e-4
    assert (db.cpu().double() - br.grad).abs().max() < tol * rows ** 0.5 * 4 + 1e-4


def attn_ref(qq, kk, vv, H):
    B, NQ, C = qq.shape
    NK = kk.shape[1]
    hd = C // H
    qh = qq.view(B, NQ, H, hd).transpose(1, 2)
    kh = kk.view(B, NK, H, hd).transpose(1, 2)
    vh = vv.view(B, NK, H, hd).transpose(1, 2)
    att = F.softmax(qh @ kh.transpose(-2, -1) / math.sqrt(hd), dim=-1)
    return (att @ vh).transpose(1, 2).reshape(B, NQ, C)


@pytest.mark.parametrize("dtype,tol,generic", [(_lib.F32, 2e-5, 1), (_lib.BF16, 2e-2, 0)])
@pytest.mark.parametrize("B,H,NQ,NK", [(1, 2, 256, 2048), (1, 2, 256, 7936), (1, 1, 256, 8448), (1, 1, 8192, 256), (1, 1, 4096, 4352), (2, 1, 8192, 8192)])
def test_attention_fwd_bwd_block8192(dtype, tol, generic, B, H, NQ, NK):
    """The attention shapes of the 128-frame configs (block 8192; VERDICT r03 weak #8: the backward had never run beyond 769 keys):
    latent_enc with 2048 / 7936 keys, lt2l with 256 + 8192 keys, latent_dec with 8192 queries, the t = 0.5 training shapes
    (4096 queries, 4352 keys) and the all-`maskgit` geometry (8192 x 8192), forward and both backward kernels against fp64 torch."""
    test_attention_fwd_bwd(dtype, tol, generic, B, H, NQ, NK, 64)


@pytest.mark.parametrize("dtype,tol,generic", [(_lib.F32, 2e-5, 1), (_lib.BF16, 2e-2, 1), (_lib.BF16, 2e-2, 0)])
@pytest.mark.parametrize("B,H,NQ,NK,HD", [(2, 2, 70, 45, 32), (2, 4, 64, 130, 64), (1, 3, 200, 1, 64), (2, 2, 33, 0, 64), (1, 2, 256, 513, 64),
                                         (1, 2, 129, 257, 64), (1, 1, 300, 769, 64), (1, 2, 385, 256, 64),
                                         # the launcher's forward forms by grid size (attention_mfma.hip launch_attn_fwd_mfma): ragged long key sets on a
                                         # small grid (split-keys form), ragged shapes on grids of more than 128 workgroups (8-wave forms, 1 / 2 stages)
                                         (1, 2, 200, 3001, 64), (2, 1, 64, 1025, 64), (5, 16, 300, 769, 64), (9, 16, 128, 200, 64)])
def test_attention_fwd_bwd(dtype, tol, generic, B, H, NQ, NK, HD):
    t = tdt(dtype)
    C = H * HD
    # q/k/v packed like the engine packs them: q in its own buffer, k|v interleaved per row
    qq = q(rnd(B, NQ, C, seed=1), dtype)
    kv = q(rnd(B, NK, 2 * C, seed=2), dtype)
    do = q(rnd(B, NQ, C, seed=3), dtype)
    qr, kvr = qq.double().requires_grad_(True), kv.double().requires_grad_(True)
    if NK > 0:
        ref = attn_ref(qr, kvr[..., :C], kvr[..., C:], H)
        (ref * do.double()).sum().backward()
    else:
        ref = torch.zeros(B, NQ, C, dtype=torch.float64)
    qd, kvd = qq.to(DEV, t), kv.to(DEV, t)
    o = torch.full((B, NQ, C), float("nan"), device=DEV, dtype=t)
    lse = torch.empty(B, H, NQ, device=DEV)
    kp = ptr(kvd) if NK > 0 else None
    vp = kvd.data_ptr() + C * kvd.element_size() if NK > 0 else None
    check(lib().mebt_op_attention_fwd(dtype, ptr(qd), kp, vp, ptr(o), ptr(lse), B, H, NQ, NK, HD, C, 2 * C, 2 * C, C, generic, cur_stream()))
    torch.cuda.synchronize()
    assert (o.float().cpu().double() - ref.detach()).abs().max() < tol
    dq = torch.full((B, NQ, C), float("nan"), device=DEV, dtype=t)
    dkv = torch.full((B, NK, 2 * C), float("nan"), device=DEV, dtype=t)
    delta = torch.empty(B, H, NQ, device=DEV)
    dkp = ptr(dkv) if NK > 0 else None
    dvp = dkv.data_ptr() + C * dkv.element_size() if NK > 0 else None
    dod = do.to(DEV, t)
    check(lib().mebt_op_attention_bwd(dtype, ptr(qd), kp, vp, ptr(o), ptr(lse), ptr(dod), ptr(dq), dkp, dvp, ptr(delta),
                                      B, H, NQ, NK, HD, C, 2 * C, 2 * C, C, generic, cur_stream()))
    torch.cuda.synchronize()
    if NK > 0:
        assert (dq.float().cpu().double() - qr.grad).abs().max() < tol * 4 * max(1.0, qr.grad.abs().max().item())
        assert (dkv.float().cpu().double() - kvr.grad).abs().max() < tol * 4 * max(1.0, kvr.grad.abs().max().item())
    else:
        assert dq.float().abs().max() == 0


def test_embed_matches_oracle():
    from oracle import mebt_oracle as orc
    cfg = orc.OracleConfig(1, 2, 64, 48, 8, ["latent_enc"], shape=(3, 4, 4))
    P = orc.closed_form_params(cfg)
    B, N, NC = 3, 48, 17
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 16384, (B, N), generator=g)
    idx = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
    ci, ti = idx[:, :NC].contiguous(), idx[:, NC:].contiguous()
    sos_r, ctx_r, tgt_r = orc.embed(P, cfg, x, ci, ti)
    for dtype in (_lib.F32, _lib.BF16):
        t = tdt(dtype)
        sos = torch.empty(B, 8, 64, device=DEV, dtype=t)
        ctx = torch.empty(B, NC, 64, device=DEV, dtype=t)
        tgt = torch.empty(B, N - NC, 64, device=DEV, dtype=t)
        dev = {k: v.to(DEV) for k, v in P.items()}
        xd, cid, tid = x.to(DEV), ci.to(DEV), ti.to(DEV)
        check(lib().mebt_op_embed_fwd(dtype, ptr(xd), ptr(cid), ptr(tid), ptr(dev["tok_emb.weight"]),
                                      ptr(dev["pos_emb"]), ptr(dev["mask_emb"]), ptr(dev["sos_emb"]), ptr(sos), ptr(ctx), ptr(tgt),
                                      B, N, NC, N - NC, 8, 64, 16384, 48, cur_stream()))
        torch.cuda.synchronize()
        for got, ref in ((sos, sos_r), (ctx, ctx_r), (tgt, tgt_r)):
            if dtype == _lib.F32:
                assert torch.equal(got.cpu(), ref.contiguous())          # pure copies/adds: bit-exact
            else:
                assert torch.equal(got.cpu(), ref.to(torch.bfloat16))    # one rounding


def test_sampler_ops_match_oracle_and_golden():
    import os
    from oracle import mebt_oracle as orc
    from oracle import closed_form as cf
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sampler_ops.npz"))
    logits = torch.from_numpy(g["logits"])
    R, V = logits.shape[0] * logits.shape[1], logits.shape[2]
    for i, (temp, k, p) in enumerate(g["cases"]):
        noise = torch.from_numpy(cf.exp1_noise("noise", tuple(logits.shape), stream=int(g[f"s{i}_stream"])))
        ids = torch.empty(R, dtype=torch.long, device=DEV)
        score = torch.empty(R, device=DEV)
        probs = torch.empty(R, V, device=DEV)
        ld, nd = logits.to(DEV), noise.to(DEV)
        check(lib().mebt_op_sample(ptr(ld), ptr(nd), float(temp), int(k), float(p), ptr(ids), ptr(score),
                                   ptr(probs), R, V, cur_stream()))
        torch.cuda.synchronize()
        assert (ids.cpu().numpy().reshape(g[f"s{i}_ids"].shape) == g[f"s{i}_ids"]).all(), i
        np.testing.assert_allclose(probs.cpu().numpy().reshape(g[f"s{i}_probs"].shape), g[f"s{i}_probs"], atol=2e-7, rtol=2e-5)
        ref_score = torch.from_numpy(g[f"s{i}_probs"]).reshape(R, V).gather(1, torch.from_numpy(g[f"s{i}_ids"]).reshape(R, 1)).squeeze(1)
        np.testing.assert_allclose(score.cpu().numpy(), ref_score.numpy(), rtol=2e-5, atol=1e-8)
    # scatter + next mask
    ci, ti, score = (torch.from_numpy(g[k]) for k in ("g_ci", "g_ti", "g_score"))
    B, NC = ci.shape
    NT = ti.shape[1]
    for i, (strategy, (ctemp, nm)) in enumerate(zip(g["g_strategy"], g["g_cases"])):
        if strategy in ("random", "bootstrap"):
            continue   # host logic swaps the score for randn before calling the kernel (tested at model level)
        n_ctx = NC + NT - int(nm)
        if n_ctx <= NC:
            continue
        n_new = n_ctx - NC
        noise = torch.from_numpy(cf.exp1_noise("noise", tuple(score.shape), stream=int(g[f"g{i}_stream"])))
        nc = torch.empty(B, NC + n_new, dtype=torch.long, device=DEV)
        nt = torch.empty(B, NT - n_new, dtype=torch.long, device=DEV)
        cid, tid, sd, nd = ci.to(DEV), ti.to(DEV), score.to(DEV), noise.to(DEV)
        check(lib().mebt_op_next_mask(ptr(cid), ptr(tid), ptr(sd), ptr(nd), float(ctemp), n_new,
                                      B, NC, NT, ptr(nc), ptr(nt), cur_stream()))
        torch.cuda.synchronize()
        assert (nc.cpu().numpy() == g[f"g{i}_ctx"]).all() and (nt.cpu().numpy() == g[f"g{i}_tgt"]).all(), (i, strategy)
    x = torch.zeros(B, 32, dtype=torch.long, device=DEV)
    ids = torch.arange(B * NT, device=DEV).view(B, NT) + 100
    tid = ti.to(DEV)
    check(lib().mebt_op_scatter_ids(ptr(x), ptr(tid), ptr(ids), B, 32, NT, cur_stream()))
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), orc.scatter_ids(torch.zeros(B, 32, dtype=torch.long), ti, ids.cpu()))


def test_sampler_full_vocab_row():
    """Full 16384-entry rows incl. top-k radix select and top-p sort path, vs the oracle."""
    from oracle import mebt_oracle as orc
    R, V = 6, 16384
    logits = rnd(R, V, seed=11, scale=1.5)
    noise = torch.empty(R, V).exponential_(generator=torch.Generator().manual_seed(5))
    for temp, k, p in ((1.0, 0, 0.0), (0.9, 32, 0.0), (1.0, 0, 0.9), (0.7, 100, 0.8)):
        ids_r, probs_r = orc.sample_from_logits(logits, temp, k or None, p or None, noise)
        ids = torch.empty(R, dtype=torch.long, device=DEV)
        score = torch.empty(R, device=DEV)
        probs = torch.empty(R, V, device=DEV)
        ld, nd = logits.to(DEV), noise.to(DEV)
        check(lib().mebt_op_sample(ptr(ld), ptr(nd), temp, k, p, ptr(ids), ptr(score), ptr(probs), R, V, cur_stream()))
        torch.cuda.synchronize()
        assert torch.equal(ids.cpu(), ids_r), (temp, k, p)
        assert ((probs.cpu() > 0) == (probs_r > 0)).all()
        np.testing.assert_allclose(probs.cpu().numpy(), probs_r.numpy(), rtol=3e-5, atol=1e-9)


def test_sampler_register_kernel_edges_and_scattered_probability_map():
    """The register-resident sampler (V = 16384, no top-p: csrc/sampler.hip sample_fast_kernel) on the rows that leave its fast
    top-k path: k = 1, k = 256 (the last k served by the local-maxima bound), k = 300 and k = V - 1 (radix fallback), a row of
    equal values and a row with two distinct values (more than 1024 candidates -> radix fallback), rows that are -inf except for a
    few entries, NaNs, temperature != 1 — ids and probabilities against the oracle; then the debug=True form: probabilities
    written straight to rows ti of a [B, N, V] map == materialise + scatter_ (transformer.py:426-436)."""
    from oracle import mebt_oracle as orc
    V = 16384
    g = torch.Generator().manual_seed(21)
    rows = [torch.randn(V, generator=g) * 2.0 for _ in range(4)]
    rows.append(torch.zeros(V))                                              # all equal
    r = torch.zeros(V); r[::3] = 1.0; rows.append(r)                         # two values, thousands of ties at the threshold
    r = torch.full((V,), -float("inf")); r[[5, 900, 16383]] = torch.tensor([0.3, 1.2, -0.7]); rows.append(r)
    r = torch.randn(V, generator=g); r[100:200] = float("nan"); rows.append(r)
    r = torch.sort(torch.randn(V, generator=g), descending=True).values; rows.append(r)      # the whole top-k inside one thread's stride
    r = torch.randn(V, generator=g); r[:256 * 40:256] = 7.0; rows.append(r)                   # 40 equal maxima owned by ONE thread
    logits = torch.stack(rows)
    R = logits.shape[0]
    noise = torch.empty(R, V).exponential_(generator=g)
    ld, nd = logits.to(DEV), noise.to(DEV)
    for temp, k in ((1.0, 0), (1.0, 1), (1.0, 32), (0.7, 32), (1.0, 256), (1.0, 300), (1.3, 16383), (0.05, 5)):
        ids_r, probs_r = orc.sample_from_logits(logits, temp, k or None, None, noise)
        ids = torch.empty(R, dtype=torch.long, device=DEV)
        score = torch.empty(R, device=DEV)
        probs = torch.full((R, V), -7.0, device=DEV)
        check(lib().mebt_op_sample(ptr(ld), ptr(nd), temp, k, 0.0, ptr(ids), ptr(score), ptr(probs), R, V, cur_stream()))
        torch.cuda.synchronize()
        assert torch.equal(ids.cpu(), ids_r), (temp, k, ids.cpu().tolist(), ids_r.tolist())
        # the kept sets agree; fp32 denormals (exp of a spread beyond 87 at temperature 0.05) are flushed to zero on the GPU, kept by the CPU
        assert ((probs.cpu() > 0) == (probs_r > 0))[probs_r > 1e-30].all() and not (probs.cpu() > 0)[probs_r == 0].any(), (temp, k)
        np.testing.assert_allclose(probs.cpu().numpy(), probs_r.numpy(), rtol=3e-5, atol=1e-9)
        np.testing.assert_allclose(score.cpu().numpy(), probs_r.gather(1, ids_r.view(R, 1)).squeeze(1).numpy(), rtol=3e-5, atol=1e-12)
    # the scattered probability map
    B, N, NT = 2, 9, 5
    lg = torch.randn(B * NT, V, generator=g)
    nz = torch.empty(B * NT, V).exponential_(generator=g)
    ti = torch.stack([torch.randperm(N, generator=g)[:NT] for _ in range(B)])
    ids_r, probs_r = orc.sample_from_logits(lg, 0.9, 32, None, nz)
    ref_map = -torch.ones(B, N, V)
    ref_map.scatter_(1, ti.unsqueeze(-1).expand(-1, -1, V), probs_r.view(B, NT, V))
    pmap = -torch.ones(B, N, V, device=DEV)
    ids = torch.empty(B * NT, dtype=torch.long, device=DEV)
    score = torch.empty(B * NT, device=DEV)
    lgd, nzd, tid = lg.to(DEV), nz.to(DEV), ti.to(DEV)
    check(lib().mebt_op_sample_scatter(ptr(lgd), ptr(nzd), 0, 0.9, 32, ptr(ids), ptr(score), ptr(pmap), ptr(tid), B, N, NT, V, cur_stream()))
    torch.cuda.synchronize()
    assert torch.equal(ids.cpu(), ids_r)
    assert ((pmap.cpu() == -1) == (ref_map == -1)).all()
    np.testing.assert_allclose(pmap.cpu().numpy(), ref_map.numpy(), rtol=3e-5, atol=1e-9)


def test_sampler_bf16_logits_same_draw_as_fp32_kernel_on_the_rounded_values():
    """`mebt_op_sample_lp` on bf16 logits (what the head of a bf16 model hands to the draw in the sampling loops: mebt_forward flag 4)
    against the oracle and the fp32 kernel on the SAME values converted back to fp32: ids, scores and the scattered probability map
    identical — the bf16 path changes where the logits are rounded, not the draw (reference transformer.py:826-889 on those values)."""
    from oracle import mebt_oracle as orc
    V, B, N, NT = 16384, 2, 11, 6
    g = torch.Generator().manual_seed(77)
    lg32 = torch.randn(B * NT, V, generator=g) * 2.5
    lg32[3, 40:60] = float("-inf")
    lb = lg32.bfloat16()
    back = lb.float()
    nz = torch.empty(B * NT, V).exponential_(generator=g)
    ti = torch.stack([torch.randperm(N, generator=g)[:NT] for _ in range(B)])
    lbd, backd, nzd, tid = lb.to(DEV), back.to(DEV), nz.to(DEV), ti.to(DEV)
    for temp, k in ((1.0, 0), (0.9, 32), (1.0, 600)):
        ids_r, probs_r = orc.sample_from_logits(back, temp, k or None, None, nz)
        out = {}
        for name in ("lp", "f32"):
            ids = torch.empty(B * NT, dtype=torch.long, device=DEV)
            score = torch.empty(B * NT, device=DEV)
            pmap = -torch.ones(B, N, V, device=DEV)
            if name == "lp":
                check(lib().mebt_op_sample_lp(ptr(lbd), 1, ptr(nzd), 0, temp, k, ptr(ids), ptr(score), ptr(pmap), ptr(tid), B, N, NT, V, 0, cur_stream()))
            else:
                check(lib().mebt_op_sample_scatter(ptr(backd), ptr(nzd), 0, temp, k, ptr(ids), ptr(score), ptr(pmap), ptr(tid), B, N, NT, V, cur_stream()))
            torch.cuda.synchronize()
            out[name] = (ids.cpu(), score.cpu(), pmap.cpu())
        assert torch.equal(out["lp"][0], ids_r) and torch.equal(out["lp"][0], out["f32"][0]), (temp, k)
        assert torch.equal(out["lp"][1], out["f32"][1]) and torch.equal(out["lp"][2], out["f32"][2]), (temp, k)
        ref_map = -torch.ones(B, N, V)
        ref_map.scatter_(1, ti.unsqueeze(-1).expand(-1, -1, V), probs_r.view(B, NT, V))
        np.testing.assert_allclose(out["lp"][2].numpy(), ref_map.numpy(), rtol=3e-5, atol=1e-9)
    # without a map, drawn from the in-kernel generator: same ids as the fp32 kernel on the same values and seed
    ids_a = torch.empty(B * NT, dtype=torch.long, device=DEV)
    ids_b = torch.empty(B * NT, dtype=torch.long, device=DEV)
    sc_a, sc_b = torch.empty(B * NT, device=DEV), torch.empty(B * NT, device=DEV)
    check(lib().mebt_op_sample_lp(ptr(lbd), 1, None, 1234567, 1.0, 32, ptr(ids_a), ptr(sc_a), None, None, B, NT, NT, V, 0, cur_stream()))
    check(lib().mebt_op_sample_seeded(ptr(backd), 1234567, 1.0, 32, 0.0, ptr(ids_b), ptr(sc_b), None, B * NT, V, cur_stream()))
    torch.cuda.synchronize()
    assert torch.equal(ids_a, ids_b) and torch.equal(sc_a, sc_b)
    # fp32 logits through the same entry point
    check(lib().mebt_op_sample_lp(ptr(backd), 0, None, 1234567, 1.0, 32, ptr(ids_a), ptr(sc_a), None, None, B, NT, NT, V, 0, cur_stream()))
    torch.cuda.synchronize()
    assert torch.equal(ids_a, ids_b)
    with pytest.raises(Exception):      # another vocabulary: the register kernel only
        check(lib().mebt_op_sample_lp(ptr(lbd), 1, None, 1, 1.0, 0, ptr(ids_a), ptr(sc_a), None, None, B, NT, NT, 8192, 0, cur_stream()))


def test_sampler_inverse_cdf_draw_is_a_sample_of_the_same_distribution():
    """The production draw of the sampling loops (mebt_op_sample_lp, draw = 1; no injected noise): ONE uniform per row keyed by
    (seed, row), the first element — in the kernel's order: thread 0's four runs of eight elements, thread 1's, ... — whose running sum of p
    reaches u * sum(p).  The reference's draw (arg-max p / q, q ~ Exp(1), transformer.py:826-841) is a sample of the categorical
    distribution p; so is this.  Checked: (1) against a float64 inverse CDF in the kernel's element order driven by the CPU twin of
    the uniform (oracle/closed_form.py:uniform_counter): the same element, or — where fp32 rounding of the running sum can move the
    crossing — a neighbour in that order among the positive-probability elements; (2) score = p[id] of the fp32 kernel's
    probabilities; (3) top-k: never an element the filter removed; (4) one-hot rows; (5) the empirical distribution over 16384 rows
    of identical logits (chi-square); (6) determinism per (seed, row)."""
    from oracle import closed_form as cf
    from oracle import mebt_oracle as orc
    V, R = 16384, 64
    g = torch.Generator().manual_seed(91)
    logits = torch.randn(R, V, generator=g) * 2.5
    logits[5] = -float("inf"); logits[5, 777] = 0.3                         # one-hot
    logits[6, 100:9000] = -float("inf")
    ld = logits.to(DEV)
    seed = 0x51ED_0000_1234
    # kernel order: thread t owns elements 8 t + k + 4096 g (g = 0 .. 3 major, k = 0 .. 7), threads in order
    order = orc.inverse_cdf_order(V)
    for temp, k in ((1.0, 0), (0.8, 32), (1.0, 700)):
        ids = torch.empty(R, dtype=torch.long, device=DEV)
        score = torch.empty(R, device=DEV)
        check(lib().mebt_op_sample_lp(ptr(ld), 0, None, seed, temp, k, ptr(ids), ptr(score), None, None, 1, R, R, V, 1, cur_stream()))
        torch.cuda.synchronize()
        ids, score = ids.cpu(), score.cpu()
        # the oracle's twin of the draw (oracle/mebt_oracle.py:sample_inverse_cdf: same uniform, same element order, the oracle's own
        # post-top-k probabilities, float64 running sum): the same element, or - where fp32 rounding of the running sum can move the
        # crossing (margin = distance of u * sum(p) from the chosen element's boundaries) - its neighbour among the positive elements
        oid, probs, margin = orc.sample_inverse_cdf(logits, temp, k or None, seed)
        po = probs.double()[:, order]
        for r in (ids != oid).nonzero().flatten().tolist():
            pos_idx = (po[r] > 0).nonzero().flatten()
            got, exact = int((order == ids[r]).nonzero().flatten()[0]), int((order == oid[r]).nonzero().flatten()[0])
            assert po[r, got] > 0, (temp, k, r)
            a, b = int((pos_idx == exact).nonzero().flatten()[0]), int((pos_idx == got).nonzero().flatten()[0])
            assert abs(a - b) == 1 and float(margin[r]) < 2e-5, (temp, k, r, exact, got, float(margin[r]))
        assert int((ids != oid).sum()) <= 2, (temp, k)
        np.testing.assert_allclose(score.numpy(), probs.gather(1, ids.view(R, 1)).squeeze(1).numpy(), rtol=3e-5, atol=1e-12)
        assert int(ids[5]) == 777
        ids2 = torch.empty(R, dtype=torch.long, device=DEV)
        check(lib().mebt_op_sample_lp(ptr(ld), 0, None, seed, temp, k, ptr(ids2), None, None, None, 1, R, R, V, 1, cur_stream()))
        assert torch.equal(ids2.cpu(), ids)
        check(lib().mebt_op_sample_lp(ptr(ld), 0, None, seed + 1, temp, k, ptr(ids2), None, None, None, 1, R, R, V, 1, cur_stream()))
        assert (ids2.cpu() != ids).float().mean().item() > 0.4
        # bf16 logits: the same draw on the rounded values
        lb = logits.bfloat16()
        idb, idf = torch.empty(R, dtype=torch.long, device=DEV), torch.empty(R, dtype=torch.long, device=DEV)
        lbd, lbf = lb.to(DEV), lb.float().to(DEV)
        check(lib().mebt_op_sample_lp(ptr(lbd), 1, None, seed, temp, k, ptr(idb), None, None, None, 1, R, R, V, 1, cur_stream()))
        check(lib().mebt_op_sample_lp(ptr(lbf), 0, None, seed, temp, k, ptr(idf), None, None, None, 1, R, R, V, 1, cur_stream()))
        assert torch.equal(idb, idf)
    # distribution: 16384 rows of the SAME logits, 12 elements carry the mass
    row = torch.full((V,), -30.0)
    sup = torch.tensor([3, 511, 512, 1000, 4097, 8191, 8192, 12000, 16000, 16383, 700, 701])
    row[sup] = torch.tensor([2.0, 1.5, 1.0, 0.5, 0.0, -0.5, 1.2, 0.3, -1.0, 0.8, 0.1, 1.7])
    Rn = 16384
    many = row.repeat(Rn, 1).to(DEV)
    ids = torch.empty(Rn, dtype=torch.long, device=DEV)
    check(lib().mebt_op_sample_lp(ptr(many), 0, None, 987654321, 1.0, 0, ptr(ids), None, None, None, 1, Rn, Rn, V, 1, cur_stream()))
    torch.cuda.synchronize()
    p = torch.softmax(row.double(), 0)
    counts = torch.bincount(ids.cpu(), minlength=V).double()
    assert counts[sup].sum() >= Rn - 2                                      # the rest of the row carries 2e-10 of the mass
    chi2 = float((((counts[sup] - Rn * p[sup]) ** 2) / (Rn * p[sup])).sum())
    assert chi2 < 40.0, chi2                                                # 11 degrees of freedom: P(chi2 > 40) = 4e-5
    with pytest.raises(Exception):      # the inverse-CDF draw takes no noise tensor
        nz = torch.ones(R, V, device=DEV)
        check(lib().mebt_op_sample_lp(ptr(ld), 0, ptr(nz), 1, 1.0, 0, ptr(ids2), None, None, None, 1, R, R, V, 1, cur_stream()))


@pytest.mark.parametrize("NT,NC,ctemp", [(8192, 0, 2.0), (8128, 64, 0.0), (5000, 3192, 1.3), (33, 7, 4.5)])
def test_next_mask_kernel_long_rows(NT, NC, ctemp):
    """generate_next_mask at the 128-frame geometry (up to 8192 targets per row; 1024-thread LDS bitonic sort) against the oracle's
    argsort: the same keys in fp32, stable by index at exact ties."""
    B = 3
    g = torch.Generator().manual_seed(NT)
    perm = torch.stack([torch.randperm(NC + NT, generator=g) for _ in range(B)])
    ci, ti = perm[:, :NC].contiguous(), perm[:, NC:].contiguous()
    score = torch.rand(B, NT, generator=g)
    score[0, : NT // 2] = score[0, NT // 2: 2 * (NT // 2)]                   # exact ties: resolved by position
    noise = torch.empty(B, NT).exponential_(generator=g)
    n_new = max(1, NT // 3)
    nc = torch.empty(B, NC + n_new, dtype=torch.long, device=DEV)
    nt = torch.empty(B, NT - n_new, dtype=torch.long, device=DEV)
    cid, tid, sd, nd = ci.to(DEV), ti.to(DEV), score.to(DEV), noise.to(DEV)
    check(lib().mebt_op_next_mask(ptr(cid) if NC else None, ptr(tid), ptr(sd), ptr(nd), float(ctemp), n_new, B, NC, NT, ptr(nc), ptr(nt), cur_stream()))
    torch.cuda.synchronize()
    nc, nt = nc.cpu(), nt.cpu()
    assert torch.equal(nc[:, :NC], ci)
    for b in range(B):
        order = torch.cat([nc[b, NC:], nt[b]])
        assert sorted(order.tolist()) == sorted(ti[b].tolist())
        pos = {int(v): j for j, v in enumerate(ti[b].tolist())}
        js = torch.tensor([pos[int(v)] for v in order])
        key = (score[b] / score[b].sum()).double() / (noise[b].double() ** ctemp if ctemp else torch.ones(NT, dtype=torch.float64))
        ko = key[js]
        assert bool((ko[1:] <= ko[:-1] * (1 + 1e-5)).all())                 # descending in the oracle's keys up to fp32 rounding of the key
        same = ko[1:] == ko[:-1]
        assert bool((js[1:][same] > js[:-1][same]).all())                   # exact ties keep their position order


@pytest.mark.parametrize("top_k,top_p", [(0, 0.0), (64, 0.0), (0, 0.9), (256, 0.8)])
def test_sampler_seeded_noise_matches_cpu_twin(top_k, top_p):
    """mebt_op_sample_seeded: the Exp(1) noise is generated inside the kernel (counter-based); with the CPU twin of the
    generator (oracle/closed_form.py:exp1_counter) the oracle must draw the same ids, except where its two best keys p/q are
    within 1e-4 of each other (the kernel's logf and the twin's float64 log differ in the last bit)."""
    from oracle import closed_form as cf
    from oracle import mebt_oracle as orc
    g = torch.Generator().manual_seed(11)
    R, V = 96, 16384
    logits = torch.randn(R, V, generator=g) * 2.0
    seed = 0x1234_5678_9ABC
    ids = torch.empty(R, dtype=torch.long, device=DEV)
    score = torch.empty(R, device=DEV)
    probs = torch.empty(R, V, device=DEV)
    lgd = logits.to(DEV)
    check(lib().mebt_op_sample_seeded(ptr(lgd), seed, 0.9, top_k, top_p, ptr(ids), ptr(score), ptr(probs), R, V, cur_stream()))
    noise = torch.from_numpy(cf.exp1_counter(seed, R, V))
    assert abs(float(noise.mean()) - 1.0) < 0.01 and float(noise.min()) > 0
    ref_ids, ref_p = orc.sample_from_logits(logits, 0.9, top_k or None, top_p or None, noise)
    np.testing.assert_allclose(probs.cpu().numpy(), ref_p.numpy(), rtol=2e-4, atol=1e-9)
    mism = (ids.cpu() != ref_ids).nonzero().flatten().tolist()
    for r in mism:
        key = ref_p[r].double() / noise[r].double()
        top2 = key.topk(2).values
        assert top2[0] / top2[1] < 1 + 1e-4, (r, float(top2[0] / top2[1]))
    assert len(mism) <= 2
    # a different seed gives different draws; the same seed the same
    ids2 = torch.empty_like(ids)
    check(lib().mebt_op_sample_seeded(ptr(lgd), seed + 1, 0.9, top_k, top_p, ptr(ids2), None, None, R, V, cur_stream()))
    assert (ids2 != ids).float().mean().item() > 0.5
    check(lib().mebt_op_sample_seeded(ptr(lgd), seed, 0.9, top_k, top_p, ptr(ids2), None, None, R, V, cur_stream()))
    assert torch.equal(ids2, ids)


def test_top_k_logits_helper_matches_reference_semantics():
    """`top_k_logits` (transformer.py:891-895): values below the k-th largest become -inf, ties with it are kept"""
    from mebt_amd.transformer import top_k_logits
    g = torch.Generator().manual_seed(5)
    lg = torch.randn(3, 7, 16384, generator=g)
    lg[0, 0, 100] = lg[0, 0].topk(5).values[-1]                  # a tie with the 5-th value
    out = top_k_logits(lg.to(DEV), 5).cpu()
    v = lg.topk(5, dim=-1).values[..., -1:]
    ref = lg.clone()
    ref[ref < v] = -float("inf")
    assert torch.equal(out, ref) and int((out[0, 0] > -float("inf")).sum()) == 6
    # kept entries stay kept however small their softmax weight: a spread of 400 underflows exp() in fp32
    lg2 = torch.linspace(-400.0, 0.0, 16384).repeat(2, 1)
    out2 = top_k_logits(lg2.to(DEV), 16000).cpu()
    assert int((out2[0] > -float("inf")).sum()) == 16000 and torch.equal(out2[0, 384:], lg2[0, 384:])
    assert torch.equal(top_k_logits(lg2.to(DEV), 16384).cpu(), lg2)        # k = V: nothing is dropped
    with pytest.raises(RuntimeError, match="no CPU path"):
        top_k_logits(lg2, 5)


def test_top_p_probs_and_gumbel_sort_helpers():
    """module-level helpers the reference exposes (transformer.py:826-841, :898-910) against the oracle's restatement"""
    from mebt_amd.transformer import top_p_probs, gumbel_sort
    from oracle import mebt_oracle as orc
    g = torch.Generator().manual_seed(9)
    probs = torch.softmax(torch.randn(5, 16384, generator=g) * 3.0, -1)
    for p in (0.3, 0.9):
        out = top_p_probs(probs.to(DEV), p).cpu()
        ref = orc.top_p_probs(probs.clone(), p)
        kept, rkept = out > 0, ref > 0
        # the two orders of summation may disagree on the entry at which the cumulative mass crosses p: at most that one per row
        assert int((kept != rkept).sum(-1).max()) <= 1
        both = kept & rkept
        assert (out[both] / ref[both] - 1).abs().max() < 2e-3 and abs(float(out.sum(-1).min()) - 1) < 1e-5
    with pytest.raises(RuntimeError, match="no CPU path"):
        top_p_probs(probs, 0.9)
    pr = torch.rand(3, 4, 97, generator=g)
    pr[0, 0, :10] = 0.0                                             # zero-probability entries sort last
    q = torch.empty_like(pr).exponential_(generator=g)
    idx = gumbel_sort(pr.to(DEV), noise=q.to(DEV)).cpu()
    key = (pr / pr.sum(-1, keepdim=True)) / q * (pr > 0).float()
    assert torch.equal(torch.gather(key, -1, idx), key.sort(-1, descending=True)[0])
    assert set(idx[0, 0, -10:].tolist()) == set(range(10))


def _grouped_call(items, fused, step=1, lr=1e-3, wd=0.05, with_bias=True, seed=0):
    """items: [(n_out, k_in, tokens)]; returns (gW, bias grads, W, mW, vW, Wlp, refs)"""
    import ctypes as C
    g = torch.Generator().manual_seed(seed)
    n = len(items)
    offs, tot = [], 0
    for no, ki, _ in items:
        offs.append(tot)
        tot += no * ki
    dYs = [(torch.randn(t, no, generator=g) * 0.5).to(torch.bfloat16).to(DEV) for no, ki, t in items]
    Xs = [(torch.randn(t, ki, generator=g) * 0.5).to(torch.bfloat16).to(DEV) for no, ki, t in items]
    W = (torch.randn(tot, generator=g) * 0.02).to(DEV)
    mW, vW = (torch.randn(tot, generator=g) * 1e-3).to(DEV), (torch.rand(tot, generator=g) * 1e-5).to(DEV)
    gW = torch.zeros(tot, device=DEV)
    Wlp = W.to(torch.bfloat16)
    biases = [torch.zeros(no, device=DEV) for no, _, _ in items]
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    i32 = lambda vs: (C.c_int32 * n)(*vs)
    ref = {"W": W.clone(), "m": mW.clone(), "v": vW.clone()}
    check(lib().mebt_op_wgrad_grouped(n, arr(dYs), arr(Xs), i32([i[0] for i in items]), i32([i[1] for i in items]), i32([i[2] for i in items]),
                                      (C.c_int64 * n)(*offs), arr(biases) if with_bias else None, ptr(W), ptr(gW), ptr(mW), ptr(vW), ptr(Wlp),
                                      1 if fused else 0, lr, 0.9, 0.95, 1e-8, wd, step, 1.0, cur_stream()))
    torch.cuda.synchronize()
    grads = [dY.float().t() @ X.float() for dY, X in zip(dYs, Xs)]
    bsum = [dY.float().sum(0) for dY in dYs]
    return gW, biases, W, mW, vW, Wlp, offs, grads, bsum, ref


@pytest.fixture(params=[None, (256, 128, 2), (256, 128, 3), (128, 128, 2), (64, 128, 4)], ids=lambda c: "tuned" if c is None else "%dx%dr%d" % c)
def grouped_config(request):
    """every tile shape / ring depth of the grouped weight-gradient launch, incl. the 8-wave 256 x 128 form (round 6), forced through
    mebt_debug_grouped_config"""
    c = request.param
    lib().mebt_debug_grouped_config(*(c if c else (0, 0, 0)))
    yield c
    lib().mebt_debug_grouped_config(0, 0, 0)


@pytest.mark.parametrize("items", [[(256, 128, 200), (128, 512, 200), (128, 128, 77)], [(64, 64, 24), (256, 64, 24), (64, 256, 9)],
                                   [(1024, 1024, 384), (4096, 1024, 384), (1024, 4096, 384), (2048, 1024, 768)],
                                   [(1024, 4096, 448), (4096, 1024, 448), (1024, 1024, 448), (1024, 1024, 448), (2048, 1024, 200),
                                    (1024, 4096, 136), (4096, 1024, 136), (1024, 1024, 136), (3072, 1024, 136)]])
def test_wgrad_grouped_operator(items, grouped_config):
    """`mebt_op_wgrad_grouped`: the weight gradients of a block in one launch against fp32 matmuls of the same bf16 operands —
    the stored gradients, the bias gradients added up inside the launch (tile column 0, ones-fragment MFMA), and the
    optimizer-in-backward form against torch-semantics AdamW on those gradients.  Ragged tiles (64-wide weights under 128- and 256-wide
    tiles) and ragged reductions (tokens not a multiple of the 64-deep k-tile); the last case is a PAIR of blocks as the engine launches
    them since round 6 (9 products in one launch)."""
    gW, biases, W, mW, vW, Wlp, offs, grads, bsum, ref = _grouped_call(items, fused=False)
    for (no, ki, t), o, gr, b, bs in zip(items, offs, grads, biases, bsum):
        got = gW[o:o + no * ki].view(no, ki)
        assert (got - gr).abs().max().item() <= 2e-4 * max(1.0, gr.abs().max().item()), (no, ki, t)
        assert (b - bs).abs().max().item() <= 2e-4 * max(1.0, bs.abs().max().item()), (no, ki, t)
    assert torch.equal(W, ref["W"]) and torch.equal(mW, ref["m"])
    # fused AdamW, step 3: p, m, v, bf16 mirror from the same gradients (torch.optim.AdamW semantics)
    lr, wd, b1, b2, step = 1e-3, 0.05, 0.9, 0.95, 3
    gW2, biases2, W2, mW2, vW2, Wlp2, offs, grads, bsum, ref = _grouped_call(items, fused=True, step=step, lr=lr, wd=wd)
    assert float(gW2.abs().max()) == 0.0                       # the gradient is applied, not stored
    for (no, ki, t), o, gr, b, bs in zip(items, offs, grads, biases2, bsum):
        sl = slice(o, o + no * ki)
        g_ = gr.reshape(-1)
        m_ = b1 * ref["m"][sl] + (1 - b1) * g_
        v_ = b2 * ref["v"][sl] + (1 - b2) * g_ * g_
        p_ = ref["W"][sl] * (1 - lr * wd) - (lr / (1 - b1 ** step)) * m_ / (v_.sqrt() / (1 - b2 ** step) ** 0.5 + 1e-8)
        assert (mW2[sl] - m_).abs().max().item() <= 2e-4 * m_.abs().max().item()
        assert (vW2[sl] - v_).abs().max().item() <= 4e-4 * v_.abs().max().item()
        # |dp| per step is bounded by ~lr; where v is tiny the quotient amplifies the gradient's summation-order noise
        assert (W2[sl] - p_).abs().max().item() <= 2e-2 * lr, (no, ki, t, (W2[sl] - p_).abs().max().item())
        assert torch.equal(Wlp2[sl], W2[sl].to(torch.bfloat16))
        assert (b - bs).abs().max().item() <= 2e-4 * max(1.0, bs.abs().max().item())
