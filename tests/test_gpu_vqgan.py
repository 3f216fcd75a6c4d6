"""3D-VQGAN first stage on MI355X (BASELINE.json configs[4]): the HIP operators of csrc/vqgan.hip and the
shipped `mebt_amd.vqgan.VQGAN` against (i) the oracle (oracle/vqgan_oracle.py, pinned to the reference) and (ii) the golden
vectors produced by the real reference `mebt.vqgan.VQGAN` (tests/golden/vq_*.npz).  GPU only.

Tolerances: fp32 mode — activations to 1e-4 relative, token ids identical to the reference except where the REFERENCE's own
best / second-best code distances are within 1e-3 of each other (an fp32 tie: recorded in the golden file); fp16 mode
(MFMA; the codebook search itself always runs on the exact-fp32 MFMA) — stated per test."""
import argparse
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import vqgan_oracle as vq
from tests.golden import make_golden as mg

DEV = "cuda"
G = os.path.join(os.path.dirname(__file__), "golden")


def build(name, dtype):
    from mebt_amd.vqgan import VQGAN
    c = mg.VQGAN_CONFIGS[name]
    args = argparse.Namespace(n_hiddens=c["n_hiddens"], downsample=c["downsample"], image_channels=3, embedding_dim=c["embedding_dim"],
                              n_codes=c["n_codes"], sequence_length=c["video"][2], sample_every_n_frames=1, resolution=c["video"][3])
    m = VQGAN(args)
    cfg = mg.vqgan_cfg(name)
    P = vq.closed_form_params(cfg)
    m.load_state_dict(P, strict=False)
    m.compute_dtype = dtype
    return m.to(DEV).eval(), cfg, P


@pytest.mark.parametrize("kind,k,stride,cin,cout", [("conv", 3, (1, 1, 1), 32, 64), ("conv", 4, (2, 2, 2), 32, 64), ("conv", 4, (1, 2, 2), 64, 128),
                                                   ("conv", 1, (1, 1, 1), 64, 64), ("conv", 3, (1, 1, 1), 16, 24),
                                                   ("conv", 3, (1, 1, 1), 64, 64), ("conv", 3, (1, 1, 1), 128, 256),
                                                   ("convt", 4, (2, 2, 2), 64, 64), ("convt", 4, (1, 2, 2), 128, 64), ("convt", 4, (2, 2, 2), 16, 8)])
def test_conv3d_operator(kind, k, stride, cin, cout):
    """mebt_op_conv3d (direct fp32, direct fp16, MFMA fp16: the LDS-DMA kernel where Cin and Cout are multiples of 64 — 256 x 64 and
    128 x 128 tiles, two column tiles at Cout 256, 480 voxels = ragged row tiles — the first MFMA kernel otherwise) vs the oracle's
    SamePadConv3d / SamePadConvTranspose3d"""
    from mebt_amd.vqgan import VQGAN, SamePadConv3d, SamePadConvTranspose3d, _Conv
    g = torch.Generator().manual_seed(k * 100 + cin)
    B, dims = 2, (4, 6, 10)
    x = torch.randn(B, cin, *dims, generator=g)
    if kind == "conv":
        w = torch.randn(cout, cin, k, k, k, generator=g) / np.sqrt(cin * k ** 3)
        b = torch.randn(cout, generator=g) * 0.1
        ref = vq.same_pad_conv3d(x, w, b, stride)
    else:
        w = torch.randn(cin, cout, k, k, k, generator=g) / np.sqrt(cin * 8)
        b = torch.randn(cout, generator=g) * 0.1
        ref = vq.same_pad_conv_transpose3d(x, w, b, stride)
    ref_cl = ref.permute(0, 2, 3, 4, 1).contiguous()
    host = VQGAN(argparse.Namespace(n_hiddens=32, downsample=(2, 2, 2), image_channels=3, embedding_dim=32, n_codes=32))
    for dtype, mfma, tol in (("f32", False, 2e-5), ("f16", False, 4e-3), ("f16", True, 4e-3)):
        if mfma and (cin % 32 or cout % 64):
            continue
        host.compute_dtype, host.use_mfma = dtype, mfma
        cv = _Conv(w.to(DEV), b.to(DEV), (k, k, k), stride, kind == "convt", dtype)
        host._prepared = {"convs": {"t": cv}}
        xin = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, host._tdt())
        out, od = host._conv("t", xin, B, dims)
        assert tuple(out.shape) == tuple(ref_cl.shape) and od == tuple(ref.shape[2:])
        err = (out.float().cpu() - ref_cl).abs().max().item() / ref_cl.abs().max().item()
        assert err < tol, (dtype, mfma, err)
        # residual add + fp32 channels-last output
        res = torch.randn_like(ref_cl)
        out2, _ = host._conv("t", xin, B, dims, resid=res.to(DEV, host._tdt()), out_mode=1)
        assert out2.dtype == torch.float32
        err = (out2.cpu() - (ref_cl + res)).abs().max().item() / ref_cl.abs().max().item()
        assert err < tol * 1.5, (dtype, mfma, err)
    # network-boundary layouts: fp32 [B,C,T,H,W] in and out
    host.compute_dtype, host.use_mfma = "f32", False
    host._prepared = {"convs": {"t": _Conv(w.to(DEV), b.to(DEV), (k, k, k), stride, kind == "convt", "f32")}}
    o_in, _ = host._conv("t", x.to(DEV), B, dims, in_mode=1)
    assert (o_in.cpu() - ref_cl).abs().max().item() < 2e-5 * ref_cl.abs().max().item()
    o_out, _ = host._conv("t", x.permute(0, 2, 3, 4, 1).contiguous().to(DEV), B, dims, out_mode=2)
    assert tuple(o_out.shape) == tuple(ref.shape) and (o_out.cpu() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("C", [32, 64, 256])
def test_groupnorm_silu_operator(C):
    from mebt_amd import _lib
    g = torch.Generator().manual_seed(C)
    B, dims = 2, (3, 5, 7)
    x = torch.randn(B, C, *dims, generator=g) * 1.7 + 0.4
    w, b = torch.randn(C, generator=g) * 0.2 + 1, torch.randn(C, generator=g) * 0.2
    ref = vq.norm_silu(x, w, b).permute(0, 2, 3, 4, 1).contiguous()
    for code, tdt, tol in ((_lib.F32, torch.float32, 2e-5), (_lib.F16, torch.float16, 3e-3)):
        xc = x.permute(0, 2, 3, 4, 1).contiguous().to(DEV, tdt)
        y = torch.empty_like(xc)
        stats = torch.empty(B * 64, device=DEV)
        wd, bd = w.to(DEV), b.to(DEV)                 # named: a temporary would be freed (and its memory reused) before the launch
        _lib.check(_lib.load().mebt_op_groupnorm_silu(code, _lib.ptr(xc), _lib.ptr(y), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(stats),
                                                     B, int(np.prod(dims)), C, _lib.cur_stream()))
        assert (y.float().cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())


def test_codebook_argmin_operator():
    from mebt_amd import _lib
    g = torch.Generator().manual_seed(3)
    M, n_codes, d = 300, 4096, 64
    z, e = torch.randn(M, d, generator=g), torch.randn(n_codes, d, generator=g)
    dist = (z ** 2).sum(1, keepdim=True) - 2 * z @ e.t() + (e.t() ** 2).sum(0, keepdim=True)
    ref = dist.argmin(1)
    zd, ed = z.to(DEV), e.to(DEV)
    score, esq, ids = torch.empty(M, n_codes, device=DEV), torch.empty(n_codes, device=DEV), torch.empty(M, dtype=torch.long, device=DEV)
    _lib.check(_lib.load().mebt_op_codebook_argmin(_lib.ptr(zd), _lib.ptr(ed), _lib.ptr(score), _lib.ptr(esq), _lib.ptr(ids), M, n_codes, d,
                                                  _lib.cur_stream()))
    got = ids.cpu()
    top2 = torch.topk(dist, 2, dim=1, largest=False).values
    mism = got != ref
    assert ((top2[:, 1] - top2[:, 0])[mism] < 1e-3).all() and mism.sum() <= 2
    # an exact tie resolves to the first index, like torch.argmin
    e2 = e.clone()
    e2[77] = e2[5]
    z2 = e2[5:6].clone()
    z2d, e2d = z2.to(DEV), e2.to(DEV)
    _lib.check(_lib.load().mebt_op_codebook_argmin(_lib.ptr(z2d), _lib.ptr(e2d), _lib.ptr(score), _lib.ptr(esq), _lib.ptr(ids), 1,
                                                  n_codes, d, _lib.cur_stream()))
    assert int(ids[0]) == 5


@pytest.mark.parametrize("case", ["random", "near_ties", "clustered", "degenerate"])
def test_codebook_filtered_search_equals_the_exact_search(case):
    """mebt_op_codebook_argmin_filtered (approximate scores on the bf16 MFMA GEMM, exact fp32 re-evaluation of every code inside the
    rounding bound) returns the SAME ids as the exact-fp32 search on the config-5 codebook geometry (16384 codes x 256): random
    data; codes planted within 1e-6 .. 1e-3 of a query's nearest code (bf16 cannot tell them apart: the re-evaluation must); a
    clustered codebook (hundreds of candidates per row); a degenerate one (every code equal: more candidates than the list holds,
    every code is re-evaluated, the first index wins).  Reference semantics: codebook.py:52-58."""
    from mebt_amd import _lib
    g = torch.Generator().manual_seed(11)
    M, n_codes, d = 1536, 16384, 256
    z, e = torch.randn(M, d, generator=g), torch.randn(n_codes, d, generator=g)
    if case == "near_ties":
        for i, eps in enumerate((1e-6, 1e-5, 1e-4, 1e-3, 0.0)):
            e[100 + i] = z[i] + 0.05 * torch.randn(d, generator=g)               # the nearest code of query i
            e[9000 + i] = e[100 + i] + eps * torch.randn(d, generator=g)         # ... and a near-duplicate of it further down the list
    elif case == "clustered":
        centre = torch.randn(1, d, generator=g)
        e = centre + 0.01 * torch.randn(n_codes, d, generator=g)
        z = centre + 0.3 * torch.randn(M, d, generator=g)
    elif case == "degenerate":
        e = e[:1].repeat(n_codes, 1).contiguous()
    zd, ed = z.to(DEV), e.to(DEV)
    score = torch.empty(M, n_codes, device=DEV)
    esq = torch.empty(n_codes + 1, device=DEV)
    ids_x, ids_f = torch.empty(M, dtype=torch.long, device=DEV), torch.empty(M, dtype=torch.long, device=DEV)
    lowp = torch.empty((M + n_codes) * d, dtype=torch.bfloat16, device=DEV)
    lib = _lib.load()
    _lib.check(lib.mebt_op_codebook_argmin(_lib.ptr(zd), _lib.ptr(ed), _lib.ptr(score), _lib.ptr(esq), _lib.ptr(ids_x), M, n_codes, d, _lib.cur_stream()))
    _lib.check(lib.mebt_op_codebook_argmin_filtered(_lib.ptr(zd), _lib.ptr(ed), _lib.ptr(score), _lib.ptr(esq), _lib.ptr(lowp), _lib.ptr(ids_f), M, n_codes, d,
                                                    _lib.cur_stream()))
    torch.cuda.synchronize()
    a, b = ids_x.cpu(), ids_f.cpu()
    if case == "degenerate":
        assert int(b.max()) == 0 and int(a.max()) == 0
        return
    mism = (a != b).nonzero().flatten().tolist()
    # the two paths sum z . e in the same k order; a difference could only be an exact-fp32 tie broken differently
    dist = ((z ** 2).sum(1, keepdim=True) - 2 * z.double() @ e.double().t() + (e.double() ** 2).sum(1)[None]).float()
    for r in mism:
        assert abs(float(dist[r, a[r]] - dist[r, b[r]])) < 1e-4 * float(dist[r].abs().max()), (case, r, int(a[r]), int(b[r]))
    assert len(mism) <= 2, (case, len(mism))
    if case == "near_ties":
        assert [int(v) for v in b[:5]] == [int(v) for v in a[:5]] and all(int(v) in (100 + i, 9000 + i) for i, v in enumerate(b[:5]))



@pytest.mark.parametrize("name", ["vq_micro", "vq_c5"])
@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_vqgan_encode_decode_vs_reference_golden(name, dtype):
    g = np.load(os.path.join(G, name + ".npz"))
    model, cfg, P = build(name, dtype)
    x = mg.vqgan_video(name)
    emb, ids = model.encode(x.to(DEV), include_embeddings=True)
    assert ids.dtype == torch.long and tuple(ids.shape) == tuple(g["ids"].shape)
    assert tuple(emb.shape) == (x.shape[0], cfg.embedding_dim) + tuple(ids.shape[1:])
    z = model._last_z.reshape(-1, cfg.embedding_dim).cpu()
    zerr = np.abs(z[g["z_rows"]].numpy() - g["z_vals"]).max() / np.abs(g["z_vals"]).max()
    mism = ids.cpu().numpy() != g["ids"]
    gap = (g["best2"][:, 1] - g["best2"][:, 0]).reshape(g["ids"].shape)
    print(f"[vqgan {name} {dtype}] pre-VQ z rel err {zerr:.2e}; ids differing from the reference {int(mism.sum())} / {mism.size}"
          f" (largest reference gap among them {float(gap[mism].max()) if mism.any() else 0.0:.2e})")
    if dtype == "f32":
        assert zerr < 1e-4
        assert (gap[mism] < 1e-3).all() and mism.sum() <= 2           # identical up to fp32 ties of the reference itself
    else:                       # fp16 activations, fp32 search: measured z error 1.2e-3, 1 of 1024 ids (a reference gap of 4e-2 on distances ~ 300)
        assert zerr < 2.5e-3
        assert mism.mean() < 0.01 and (gap[mism] < 1e-3 * np.abs(g["best2"][:, 0]).max()).all()
    # embeddings returned with the ids are the codebook rows of those ids (straight-through value, codebook.py:60-63,94)
    e_ref = P["codebook.embeddings"][ids.cpu().reshape(-1)].view(*ids.shape, -1).permute(0, 4, 1, 2, 3)
    assert torch.equal(emb.cpu(), e_ref)
    # decode of the golden's fixed ids
    rec = model.decode(torch.from_numpy(g["dec_ids"]).to(DEV))
    assert rec.dtype == torch.float32 and tuple(rec.shape) == tuple(x.shape)
    rv = rec.reshape(-1).cpu().numpy()[g["rec_idx"]]
    rerr = np.abs(rv - g["rec_vals"]).max() / np.abs(g["rec_vals"]).max()
    print(f"[vqgan {name} {dtype}] decode rel err {rerr:.2e}")
    assert rerr < (2e-5 if dtype == "f32" else 3e-3)          # measured 3.5e-6 / 1.4e-3
    np.testing.assert_allclose(rec.mean(dim=(0, 1, 3, 4)).cpu().numpy(), g["rec_mean"], rtol=0, atol=(1e-5 if dtype == "f32" else 3e-3))
    # MFMA implicit GEMM == direct kernel (same fp16 operands, different summation order)
    if dtype == "f16":
        model.use_mfma = False
        rec2 = model.decode(torch.from_numpy(g["dec_ids"]).to(DEV))
        assert (rec2 - rec).abs().max().item() < 2e-2 * rec.abs().max().item()


def test_transformer_with_pixel_input_uses_the_first_stage():
    """`vtokens: False` (transformer.py:683-694): a pixel video goes through VQGAN.encode and gives the same logits as the
    token grid the oracle's VQGAN produces for it; draft_and_revise output decodes to a video of the input's shape."""
    from tests.helpers import product_config
    from mebt.transformer import Net2NetTransformer
    from oracle import closed_form as cf
    from oracle import mebt_oracle as orc
    vqm, vcfg, VP = build("vq_micro", "f32")
    tcfg, fscfg, mcfg = product_config("micro", vtokens=False)
    tcfg["first_stage_vocab_size"] = tcfg["vocab_size"] = 512
    model = Net2NetTransformer(tcfg, fscfg, mcfg, cond_stage_key="label")
    model.compute_dtype = "f32"
    model.first_stage_model = vqm
    ocfg = orc.OracleConfig(6, 2, 64, 32, 8, mg.CONFIGS["micro"]["mode"], vocab_size=512, shape=[2, 4, 4], budget=32, avg_loss=1.0)
    sd = {k: torch.from_numpy(v) for k, v in cf.state_dict_numpy(orc.param_shapes(ocfg)).items()}
    model.load_state_dict({k: v for k, v in sd.items()}, strict=False)
    model = model.to(DEV).eval()
    video = mg.vqgan_video("vq_micro")                       # [2, 3, 4, 16, 16] -> tokens [2, 2, 4, 4]
    with torch.no_grad():
        ids_ref = vq.encode(VP, vcfg, video)
    idx = torch.stack([torch.from_numpy(cf.permutation("vq-perm", 32, stream=b)) for b in range(2)])
    with torch.no_grad():
        lg_pix, zt_pix, _, _ = model(video.to(DEV), None, t=0.5, indices=idx.to(DEV))
        lg_tok, zt_tok, _, _ = model(ids_ref.to(DEV), None, t=0.5, indices=idx.to(DEV))
    assert torch.equal(zt_pix, zt_tok) and torch.equal(lg_pix, lg_tok)
    out = model.draft_and_revise(ids_ref.to(DEV), None, 2, 1.0, None, None, 2, 1.0, None, None, 1, False)
    rec = model.first_stage_model.decode(out.view(2, 2, 4, 4))
    assert tuple(rec.shape) == tuple(video.shape) and torch.isfinite(rec).all()
    # the script-level driver (draft_and_revise_videos.py:22-60): code map + decoded frames in [0, 1]
    from mebt_amd.sampling import draft_and_revise_sample
    log = draft_and_revise_sample(model, 2, 8, 8, 8, 2, 1.0, None, None, 2, 0.3, None, None, 2, draft=ids_ref.numpy())
    assert tuple(log["code_maps"].shape) == (2, 2, 4, 4) and tuple(log["samples"].shape) == (2, 3, 4, 16, 16)
    assert float(log["samples"].min()) >= 0.0 and float(log["samples"].max()) <= 1.0


def test_validation_epoch_hook_samples_decodes_and_logs_a_video(tmp_path):
    """`on_validation_epoch_start` (reference transformer.py:336-351): every `vis_epoch` epochs four clips are sampled from an
    all-masked grid (32 steps, cosine schedule, context temperature 6.0), decoded one by one by the first stage, clamped to
    [-0.5, 0.5] + 0.5, permuted to [N, T, C, H, W] and handed to `logger.experiment.add_video('sample', ..., epoch, fps=20)`; the
    mask schedule is restored.  Checked against the same composition spelled out by hand on the same noise."""
    from tests.helpers import product_config
    from mebt.transformer import Net2NetTransformer
    from mebt_amd.lightning_shim import VideoLogger
    from oracle import closed_form as cf
    from oracle import mebt_oracle as orc
    vqm, vcfg, VP = build("vq_micro", "f32")
    tcfg, fscfg, mcfg = product_config("micro", vtokens=False)
    tcfg["first_stage_vocab_size"] = tcfg["vocab_size"] = 512
    tcfg["vis_epoch"] = 3
    model = Net2NetTransformer(tcfg, fscfg, mcfg, cond_stage_key="label")
    model.compute_dtype = "f32"
    model.first_stage_model = vqm
    ocfg = orc.OracleConfig(6, 2, 64, 32, 8, mg.CONFIGS["micro"]["mode"], vocab_size=512, shape=[2, 4, 4], budget=32, avg_loss=1.0)
    sd = {k: torch.from_numpy(v) for k, v in cf.state_dict_numpy(orc.param_shapes(ocfg)).items()}
    model.load_state_dict({k: v for k, v in sd.items()}, strict=False)
    model = model.to(DEV).train()
    model.mask_sampler.schedule = "linear"

    class Rec:
        def __init__(self):
            self.calls, self.flushed = [], 0
            self.experiment = self

        def add_video(self, tag, vid, step, fps=4):
            self.calls.append((tag, vid.clone(), step, fps))

        def flush(self):
            self.flushed += 1

    def make_hook():
        ctr = {"k": 0}

        def hook(kind, shape):
            g = torch.Generator().manual_seed(9100 + ctr["k"])
            ctr["k"] += 1
            if kind == "perm":
                return torch.randperm(int(shape[0]), generator=g)
            if kind == "randn":
                return torch.randn(tuple(shape), generator=g)
            return torch.empty(tuple(shape), dtype=torch.float32).exponential_(generator=g)
        return hook

    rec = Rec()
    model.logger = rec
    model.current_epoch = 0                                   # (0 + 1) % 3 != 0: nothing happens
    model.on_validation_epoch_start()
    assert not rec.calls
    model.current_epoch = 2
    model.noise_hook = model.mask_sampler.noise_hook = make_hook()
    model.on_validation_epoch_start()
    assert len(rec.calls) == 1 and rec.flushed == 1
    tag, vid, step, fps = rec.calls[0]
    assert (tag, step, fps) == ("sample", 2, 20)
    assert tuple(vid.shape) == (4, 4, 3, 16, 16) and float(vid.min()) >= 0.0 and float(vid.max()) <= 1.0
    assert model.mask_sampler.schedule == "linear" and model.transformer.training      # state restored
    # the same composition by hand
    model.eval()
    model.mask_sampler.schedule = "cosine"
    model.noise_hook = model.mask_sampler.noise_hook = make_hook()
    with torch.no_grad():
        x = model.sample(torch.zeros(4, 2, 4, 4, dtype=torch.long, device=DEV), None, 1.0, None, None, 32, None, None,
                         context_temperature=6.0, skips=False)[0].reshape(4, 2, 4, 4)
        ref = torch.cat([vqm.decode(x[i:i + 1]) for i in range(4)], 0).clamp(-0.5, 0.5) + 0.5
    assert (vid - ref.permute(0, 2, 1, 3, 4)).abs().max().item() < 1e-5        # GroupNorm statistics are summed with float atomics: not bitwise run to run
    # the file logger the launcher attaches
    model.mask_sampler.schedule = "linear"
    model.logger = VideoLogger(str(tmp_path))
    model.noise_hook = model.mask_sampler.noise_hook = make_hook()
    model.on_validation_epoch_start()
    (tag, path, step, fps), = model.logger.videos
    arr = np.load(path)
    assert arr.dtype == np.uint8 and arr.shape == (4, 4, 3, 16, 16)
    assert np.abs(arr.astype(np.float32) / 255.0 - vid.cpu().numpy()).max() <= 0.5 / 255 + 2e-5      # a second sample + decode run: not bitwise (GroupNorm atomics)
    # no first stage: skipped with a warning instead of the reference's AttributeError
    model.first_stage_model = None
    with pytest.warns(UserWarning):
        model.on_validation_epoch_start()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_vqgan_other_geometry_vs_oracle(dtype):
    """Away from the two golden geometries: 3 clips of 8 frames at 96 x 64 (non-square, odd batch; n_hiddens 32, downsample
    (2,4,4), 512 codes of 64 dims) -> [3, 4, 24, 16] tokens.  Encode (pre-VQ activations, ids up to the oracle's own near
    ties) and decode against the CPU oracle, which is pinned to the reference by the golden tests above."""
    from mebt_amd.vqgan import VQGAN
    cfg = vq.VQGANConfig(32, (2, 4, 4), 3, 64, 512)
    P = vq.closed_form_params(cfg)
    m = VQGAN(argparse.Namespace(n_hiddens=32, downsample=(2, 4, 4), image_channels=3, embedding_dim=64, n_codes=512,
                                 sequence_length=8, sample_every_n_frames=1, resolution=64))
    m.load_state_dict(P, strict=False)
    m.compute_dtype = dtype
    m = m.to(DEV).eval()
    g = torch.Generator().manual_seed(17)
    x = torch.rand(3, 3, 8, 96, 64, generator=g) - 0.5
    ids_ref, z_ref, dist = vq.encode(P, cfg, x, return_all=True)
    ids = m.encode(x.to(DEV))
    assert tuple(ids.shape) == tuple(ids_ref.shape) == (3, 4, 24, 16)
    z = m._last_z.reshape(-1, 64).cpu()
    zr = z_ref.permute(0, 2, 3, 4, 1).reshape(-1, 64)
    zerr = ((z - zr).abs().max() / zr.abs().max()).item()
    best2 = torch.topk(dist, 2, dim=1, largest=False).values
    gap = (best2[:, 1] - best2[:, 0]).view(ids_ref.shape)
    mism = ids.cpu() != ids_ref
    assert zerr < (1e-4 if dtype == "f32" else 2.5e-3), zerr
    assert mism.float().mean().item() < (0.002 if dtype == "f32" else 0.02)
    if mism.any():      # only where the oracle's own two best codes are nearly tied (relative to the distances' scale)
        assert (gap[mism] < (1e-3 if dtype == "f32" else 2e-2) * best2[:, 0].abs().max()).all()
    rec_ref = vq.decode(P, cfg, ids_ref)
    rec = m.decode(ids_ref.to(DEV))
    assert tuple(rec.shape) == tuple(x.shape)
    rerr = ((rec.cpu() - rec_ref).abs().max() / rec_ref.abs().max()).item()
    print(f"[vqgan 96x64x8 {dtype}] z rel err {zerr:.2e}, ids differing {int(mism.sum())} / {mism.numel()}, decode rel err {rerr:.2e}")
    assert rerr < (2e-5 if dtype == "f32" else 3e-3)
