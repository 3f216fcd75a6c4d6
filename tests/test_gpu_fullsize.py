"""Full-size checks on MI355X: the Sky-Timelapse 16f network (BASELINE.json configs[1], 24L/1024d/16h,
337 M parameters) against the CPU oracle on the same random weights, and size-independent properties
at the UCF-128f geometry (configs[3]: block 8192, NC=7936/NT=256 revise forwards).  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mebt_amd import presets
from oracle import mebt_oracle as orc

DEV = "cuda"


def oracle_cfg_of(cfg):
    p, m = cfg.model.params, cfg.model.mask.params
    return orc.OracleConfig(p.n_layer, p.n_head, p.n_embd, p.block_size, p.sos_emb, p.mode, shape=m.shape,
                            schedule=m.schedule, budget=m.budget, avg_loss=1.0)


@pytest.fixture(scope="module")
def sky():
    torch.manual_seed(7)
    cfg = presets.sky_16f(dropout=0.0)
    model = presets.build_model(cfg, compute_dtype="f32")
    with torch.no_grad():                          # non-trivial biases / LN affine so nothing is skipped silently
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.02)
            elif ".ln" in n and n.endswith("weight"):
                p.add_(torch.randn_like(p) * 0.05)
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return cfg, model, P


def test_sky16f_logits_and_loss_fp32_within_1e3(sky):
    """north_star: logits match the reference CPU path within 1e-3 in fp32 on identical weights and
    synthetic [B,4,16,16] token grids (the oracle is the reference's algorithm, pinned by tests/golden)."""
    cfg, model, P = sky
    ocfg = oracle_cfg_of(cfg)
    g = torch.Generator().manual_seed(3)
    B = 2
    x = torch.randint(0, 16384, (B, 4, 16, 16), generator=g)
    idx = torch.stack([torch.randperm(1024, generator=g) for _ in range(B)])
    with torch.no_grad():
        ref, z_t, ntw, seq_len = orc.forward(P, ocfg, x, idx, 0.5, training=True)
        a1, a5, loss = orc.loss_and_acc(ref, z_t, ntw, seq_len, ocfg)
    model.compute_dtype = "f32"
    model.to(DEV).train()
    with torch.no_grad():
        logits, z2, ntw2, sl2 = model(x.to(DEV), None, t=0.5, indices=idx.to(DEV))
    assert logits.shape == (B, 512, 16384) and torch.equal(z2.cpu(), z_t)
    err = (logits.cpu() - ref).abs().max().item()
    assert err < 1e-3, err                                  # measured ~1e-5
    import random
    orig = random.random
    random.random = lambda: 0.5
    try:
        with torch.no_grad():
            b1, b5, l2, _ = model.shared_step({"video": x.to(DEV), "indices": idx.to(DEV)}, 0)
    finally:
        random.random = orig
    assert abs(float(l2) - float(loss)) < 1e-4 * float(loss)
    assert abs(float(b1) - float(a1)) < 0.2 and abs(float(b5) - float(a5)) < 0.2


def test_sky16f_bf16_close_to_fp32_and_training_reduces_loss():
    torch.manual_seed(11)
    cfg = presets.sky_16f(dropout=0.0)
    cfg.exp.exact_lr = 3e-4
    model = presets.build_model(cfg, compute_dtype="bf16").to(DEV).train()
    from mebt_amd.trainer import TrainLoop
    loop = TrainLoop(model)
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 64, (6, 4, 16, 16), generator=g).to(DEV)      # a learnable (low-entropy) batch
    idx = torch.stack([torch.randperm(1024, generator=g) for _ in range(6)]).to(DEV)
    losses = [float(loop.step(x, idx, t=0.5)[4].cpu()) for _ in range(12)]
    assert all(np.isfinite(losses)) and abs(losses[0] - np.log(16384)) < 0.5     # random-init CE ~ ln V
    assert losses[-1] < losses[0] - 1.0, losses                                   # the optimiser step learns


def test_sky16f_memorises_a_structured_batch_under_the_real_training_regime():
    """End to end at the benchmarked size, the way training actually runs: bf16, dropout 0.1, a fresh permutation and
    t ~ U(0,1) every step (so NC / NT and with them every GEMM / attention shape change from step to step), AdamW inside the
    weight-gradient launches.  Six "videos" whose frames repeat one 16 x 16 palette pattern are memorised: masked-token
    loss from ln(16384) = 9.7 to < 0.5 within 400 steps (measured: 3.9 / 1.9 / 0.67 / 0.08 after 50 / 200 / 300 / 400),
    every parameter finite."""
    import random
    from mebt_amd.trainer import TrainLoop
    cfg = presets.sky_16f(dropout=0.1)
    cfg.exp.exact_lr = 1e-4
    torch.manual_seed(3)
    model = presets.build_model(cfg, compute_dtype="bf16").to(DEV).train()
    loop = TrainLoop(model)
    assert loop.fused_optimizer
    g = torch.Generator().manual_seed(5)
    base = torch.randint(0, 32, (6, 1, 16, 16), generator=g)
    x = ((base + torch.arange(4).view(1, 4, 1, 1) * 7) % 16384).to(DEV)
    rng = random.Random(1)
    first = last = None
    for step in range(400):
        idx = torch.stack([torch.randperm(1024, generator=g) for _ in range(6)]).to(DEV)
        st = loop.step(x, idx, t=rng.random() * 0.98)
        if step == 0:
            first = float(st[4].cpu())
    last = float(st[4].cpu())
    assert abs(first - np.log(16384)) < 0.6 and np.isfinite(last) and last < 0.5, (first, last)
    assert all(bool(torch.isfinite(v).all()) for v in model.state_dict().values())
    # ... and the inference path reads the same weights (the bf16 mirror the optimizer epilogue kept current): given a
    # random quarter of each video it fills in the rest
    model.eval()
    perm = torch.stack([torch.randperm(1024, generator=g) for _ in range(6)]).to(DEV)
    ci, ti = perm[:, :256].contiguous(), perm[:, 256:].contiguous()
    with torch.no_grad():
        logits, _ = model.reconstruct_mask(x, ci, ti)
    truth = torch.gather(x.reshape(6, -1), 1, ti)
    acc = float((logits.argmax(-1) == truth).float().mean())
    assert acc > 0.9, acc
    print(f"[soak] loss {first:.3f} -> {last:.4f} in 400 steps; reconstruction of 768 masked tokens from 256: {100 * acc:.1f} % correct")


def test_ucf128f_geometry_revise_forward_properties():
    """C4: block_size 8192, grid [32,16,16]; a revise forward has NC=7936, NT=256.
    Properties: finite logits; bf16 agrees with fp32 loosely; permuting the CONTEXT order leaves the
    logits unchanged (attention over contexts is permutation invariant); permuting targets permutes rows."""
    torch.manual_seed(13)
    cfg = presets.ucf_128f()
    B, N, NT = 1, 8192, 256
    g = torch.Generator().manual_seed(9)
    x = torch.randint(0, 16384, (B, 32, 16, 16), generator=g).to(DEV)
    perm = torch.randperm(N, generator=g)
    ci, ti = perm[:N - NT].unsqueeze(0).to(DEV), perm[N - NT:].unsqueeze(0).to(DEV)
    model = presets.build_model(cfg, compute_dtype="f32").to(DEV).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    l32, _ = model.reconstruct_mask(x, ci, ti)
    assert l32.shape == (B, NT, 16384) and torch.isfinite(l32).all()
    shuffled = ci[:, torch.randperm(N - NT, generator=g).to(DEV)]
    l32b, _ = model.reconstruct_mask(x, shuffled, ti)
    assert (l32b - l32).abs().max().item() < 2e-4
    rp = torch.randperm(NT, generator=g).to(DEV)
    l32c, _ = model.reconstruct_mask(x, ci, ti[:, rp])
    assert (l32c - l32[:, rp]).abs().max().item() < 2e-4
    m16 = presets.build_model(presets.ucf_128f(), compute_dtype="bf16")
    m16.load_state_dict(sd)
    m16 = m16.to(DEV).eval()
    l16, _ = m16.reconstruct_mask(x, ci, ti)
    # bf16 vs fp32 on the same weights: <= 2x the 0.9 % of max |logits| measured against the oracle (test_gpu_benchsize.py)
    assert (l16 - l32).abs().max().item() < 1.7e-2 * l32.abs().max().item() and (l16.argmax(-1) == l32.argmax(-1)).float().mean() > 0.95


def test_sample_loop_full_size_invariants():
    """32-step MaskGIT sampling at Sky size: every step's (NC,NT) partition stays a
    partition of range(N); the final sample has no untouched position; greedy (T->0) is deterministic."""
    torch.manual_seed(17)
    cfg = presets.sky_16f(dropout=0.0)
    cfg.model.mask.params.schedule = "cosine"
    model = presets.build_model(cfg, compute_dtype="bf16").to(DEV).eval()
    x0 = torch.full((2, 4, 16, 16), -1, dtype=torch.long, device=DEV).clamp(min=0)
    out = model.sample(x0, None, 1.0, None, None, 8, None, None, context_temperature=6.0, skips=False, debug=True)
    xs, ci, ti, hist, ctx_hist, probs = out
    assert xs.shape == (2, 1024) and ti.shape[1] + ci.shape[1] == 1024
    both = torch.cat([ci, ti], 1).sort(dim=1).values
    assert torch.equal(both, torch.arange(1024, device=DEV).repeat(2, 1))
    assert (probs.sum(-1) - 1).abs().max().item() < 1e-3          # every position got a distribution written
    a = model.sample(x0, None, 0.0, None, None, 4, None, None, context_temperature=0.0, skips=False)[0]
    b = model.sample(x0, None, 0.0, None, None, 4, None, None, context_temperature=0.0, skips=False)[0]
    assert torch.equal(a, b)
    # these loops handed bf16 logits from the head to the draw kernel (mebt_forward flag 4 -> mebt_op_sample_lp): they are the fp32
    # logits of the public forward rounded once (other tile = other summation order: one bf16 ulp at most), and a draw from them
    # is the draw from those values
    idx = torch.stack([torch.randperm(1024, generator=torch.Generator().manual_seed(3 + b_)) for b_ in range(2)]).to(DEV)
    ci_, ti_ = idx[:, :384].contiguous(), idx[:, 384:].contiguous()
    xr = torch.randint(0, 16384, (2, 1024), generator=torch.Generator().manual_seed(5)).to(DEV)
    lb = model._sampling_logits(xr, ci_, ti_)
    l32 = model.reconstruct_mask(xr, ci_, ti_)[0]
    assert lb.dtype == torch.bfloat16 and l32.dtype == torch.float32 and lb.shape == l32.shape
    ulp = l32.abs().clamp(min=2.0 ** -126) * 2.0 ** -7
    assert ((lb.float() - l32).abs() <= ulp).all()
    assert (lb == l32.bfloat16()).float().mean().item() > 0.99
    assert model._sampling_logits(xr, ci_, ti_, temperature=1e-8).dtype == torch.float32     # (near-)greedy draws keep fp32 logits
    import os
    os.environ["MEBT_SAMPLE_BF16_LOGITS"] = "0"
    try:
        assert model._sampling_logits(xr, ci_, ti_).dtype == torch.float32
    finally:
        del os.environ["MEBT_SAMPLE_BF16_LOGITS"]


def test_key_value_cache_of_the_sampling_loops():
    """The sampling loops of a bf16 model keep the latent_enc blocks' key / value projections of all N positions (engine:
    mebt_forward_kvcache; `contexts` is read-only through the network, reference modules/gpt.py:187-192) and re-project only the
    positions whose token changed.  (1) a cached forward == the plain forward on the same state (same kernels on the attention side;
    the projection runs on other row counts, i.e. possibly another tile: bf16 rounding of a different summation order at most),
    first with every context position dirty, then after re-sampling part of the grid with only those positions dirty;
    (2) draft_and_revise / sample at Sky size under MEBT_KV_CACHE_CHECK=1: every context position of every forward holds the
    projection of its CURRENT token id (the per-position record of the session), and the loops re-project a small fraction of what
    the uncached loops project; (3) MEBT_KV_CACHE=0 gives the plain loops."""
    import os
    from mebt_amd.transformer import _KvSession
    torch.manual_seed(23)
    cfg = presets.sky_16f(dropout=0.0)
    cfg.model.mask.params.schedule = "cosine"
    model = presets.build_model(cfg, compute_dtype="bf16").to(DEV).eval()
    nm = model._ensure_native()
    B, N = 2, 1024
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 16384, (B, N), generator=g).to(DEV)
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).to(DEV)
    ci, ti = perm[:, :768].contiguous(), perm[:, 768:].contiguous()
    os.environ["MEBT_KV_CACHE_CHECK"] = "1"
    try:
        ses = _KvSession(nm, B, N)
        ref = model.reconstruct_mask(x, ci, ti)[0]
        got = ses.forward(x, ci, ti, None, False)
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() < 2e-2 * scale, ((got - ref).abs().max().item(), scale)
        assert (got.argmax(-1) == ref.argmax(-1)).float().mean().item() > 0.97
        # re-sample the 256 targets, rotate the sets: the old targets become context, 256 old context positions become targets
        x2 = x.clone()
        x2.scatter_(1, ti, torch.randint(0, 16384, (B, 256), generator=g).to(DEV))
        ci2 = torch.cat([ci[:, 256:], ti], 1).contiguous()
        ti2 = ci[:, :256].contiguous()
        ref2 = model.reconstruct_mask(x2, ci2, ti2)[0]
        got2 = ses.forward(x2, ci2, ti2, ti, False)                 # dirty = the re-sampled positions only
        assert ses.rows_projected == B * (768 + 256) and ses.rows_uncached == B * (768 + 768)
        assert (got2 - ref2).abs().max().item() < 2e-2 * ref2.abs().max().item()
        # the attention kernel gathering the cache rows itself == the cache rows copied into a contiguous buffer first (MEBT_KV_GATHER=0):
        # the same values in the same chunk order, bit for bit
        os.environ["MEBT_KV_GATHER"] = "0"
        try:
            got2c = ses.forward(x2, ci2, ti2, ti2[:, :0], False)
        finally:
            del os.environ["MEBT_KV_GATHER"]
        assert torch.equal(got2c, ses.forward(x2, ci2, ti2, ti2[:, :0], False))
        # a stale row is caught by the record: change a context token without declaring it dirty
        x3 = x2.clone()
        x3[0, ci2[0, 5]] = (x3[0, ci2[0, 5]] + 1) % 16384
        with pytest.raises(AssertionError, match="another token id"):
            ses.forward(x3, ci2, ti2, ti2[:, :0], False)
        # the loops (the invariant is asserted inside every forward)
        x0 = torch.zeros(B, 4, 16, 16, dtype=torch.long, device=DEV)
        out = model.draft_and_revise(x0, None, 4, 1.0, None, None, 4, 1.0, None, None, 2, False)
        assert out.shape == (B, N) and int(out.min()) >= 0 and int(out.max()) < 16384
        proj, unc = model._kv_last
        assert proj < 0.45 * unc, (proj, unc)           # draft 4 + 2 x revise 4: N + 3 * N / 4 ... rows instead of the contexts of 12 forwards
        out = model.sample(x0, None, 1.0, None, None, 8, None, None, context_temperature=4.5, skips=False)
        assert out[0].shape == (B, N)
        proj, unc = model._kv_last
        assert proj <= B * N and proj < 0.5 * unc, (proj, unc)       # every position enters the context once
        # revise only, as the shipped draft-and-revise scripts run it (--np_draft): first forward projects N - w rows, the other 2 n - 1 forwards w each
        out = model.draft_and_revise(torch.randint(0, 16384, (B, 4, 16, 16), generator=g).to(DEV), None, 8, 1.0, None, None, 8, 1.0, None, None, 2, True)
        proj, unc = model._kv_last
        assert proj == B * ((N - 128) + 15 * 128) and unc == B * 16 * (N - 128), (proj, unc)
        # a one-step draft (NC = 0, targets = every position) followed by revise passes - the draft forward projects nothing,
        # so the first revise forward must re-project its whole context (it used to hand the engine ND = N > NC and raise)
        out = model.draft_and_revise(x0, None, 1, 1.0, None, None, 4, 1.0, None, None, 2, False)
        proj, unc = model._kv_last
        assert out.shape == (B, N) and proj == B * ((N - 256) + 7 * 256) and unc == B * 8 * (N - 256), (proj, unc)
        # ... and one-step revise passes (again NC = 0) before a longer pass
        out = model.draft_and_revise(x0, None, 2, 1.0, None, None, 1, 1.0, None, None, 2, False)
        out = model.revise(out.view(B, 4, 16, 16), None, 1.0, None, None, 4)
        assert out.shape == (B, N)
        # a caller that names no dirty positions and whose previous targets do not fit the context: whole-context projection from
        # there on (`degraded`), never a stale row (the record is asserted inside every forward)
        ses2 = _KvSession(nm, B, N)
        ses2.forward(x, ci, ti, None, False)
        big_t, small_c = perm[:, :640].contiguous(), perm[:, 640:].contiguous()          # NT = 640 > the next forward's NC = 256 ...
        ses2.forward(x, small_c, big_t, None, False)
        x4 = x.clone()
        x4.scatter_(1, big_t, torch.randint(0, 16384, (B, 640), generator=g).to(DEV))
        got4 = ses2.forward(x4, perm[:, 512:768].contiguous(), perm[:, 768:].contiguous(), None, False)
        assert ses2.degraded
        ref4 = model.reconstruct_mask(x4, perm[:, 512:768].contiguous(), perm[:, 768:].contiguous())[0]
        assert (got4 - ref4).abs().max().item() < 2e-2 * ref4.abs().max().item()
        got5 = ses2.forward(x4, ci, ti, ti[:, :0], False)            # an (empty) dirty list is ignored once degraded: positions 0..639 changed
        ref5 = model.reconstruct_mask(x4, ci, ti)[0]
        assert (got5 - ref5).abs().max().item() < 2e-2 * ref5.abs().max().item()
        # the script drivers on top of `sample`: bootstrap + continuation windows with a fixed context, and extrapolate(edit=True)
        from mebt_amd.sampling import bidirect_sample, extrapolate
        log = bidirect_sample(model, B, 32, 16, 8, temperature=1.0, top_k=32, top_p=None, vid_n_steps=4, vid_c_temp=2.0, bootstrap=3)
        assert tuple(log["code_maps"].shape) == (B, 8, 16, 16) and bool(torch.isfinite(log["score"]).all())
        log = extrapolate(model, torch.randint(0, 16384, (B, 4, 16, 16), generator=g).to(DEV), 32, 16, 8, temperature=0.9, top_k=64, vid_n_steps=3, vid_c_temp=2.5)
        assert tuple(log["code_maps"].shape) == (B, 8, 16, 16)
    finally:
        del os.environ["MEBT_KV_CACHE_CHECK"]
    os.environ["MEBT_KV_CACHE"] = "0"
    try:
        model._kv_last = None
        model.sample(x0, None, 1.0, None, None, 4, None, None, context_temperature=4.5, skips=False)
        assert model._kv_last is None
    finally:
        del os.environ["MEBT_KV_CACHE"]
