"""The oracle (oracle/mebt_oracle.py) against the golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only; this is the pin that makes the oracle trustworthy."""
import os
import numpy as np
import pytest
import torch

from oracle import mebt_oracle as orc
from oracle import closed_form as cf
from tests.golden import make_golden as mg

G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-5      # fp32 CPU vs fp32 CPU, different op order only


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


_params = {}


def params(name):
    if name not in _params:
        _params[name] = orc.closed_form_params(mg.oracle_cfg(name))
    return _params[name]


def check_digest(logits, g, pre):
    lg = logits.double()
    np.testing.assert_allclose(torch.logsumexp(lg, -1).numpy(), g[pre + "lse"], atol=TOL, rtol=0)
    np.testing.assert_allclose(logits[..., g["cols"]].numpy(), g[pre + "cols"], atol=TOL, rtol=0)
    np.testing.assert_allclose(lg.mean(-1).numpy(), g[pre + "mean"], atol=TOL, rtol=0)
    np.testing.assert_allclose((lg * lg).sum(-1).numpy(), g[pre + "sqsum"], rtol=1e-5)
    v, i = logits.topk(5, -1)
    np.testing.assert_allclose(v.numpy(), g[pre + "top5_vals"], atol=TOL, rtol=0)
    assert (logits.argmax(-1).numpy() == g[pre + "argmax"]).mean() > 0.995   # near-ties may flip


@pytest.mark.parametrize("name", ["c1", "micro", "micro_budget", "micro_maskgit"])
def test_forward_and_loss(name):
    g = load("forward_" + name)
    cfg, P = mg.oracle_cfg(name), params(name)
    x, idx = torch.from_numpy(g["x"]), torch.from_numpy(g["indices"])
    for c, (mode, t) in enumerate(zip(g["case_mode"], g["case_t"])):
        with torch.no_grad():
            logits, z_t, ntw, seq_len = orc.forward(P, cfg, x, idx, float(t), training=(mode == "train"))
            a1, a5, loss = orc.loss_and_acc(logits, z_t, ntw, seq_len, cfg)
        meta = g[f"c{c}_meta"]
        assert (z_t.numpy() == g[f"c{c}_z_targets"]).all()
        assert ntw == meta[0] and seq_len == meta[1]
        check_digest(logits, g, f"c{c}_")
        assert abs(float(loss) - meta[2]) < 1e-5 * max(1, abs(meta[2]))
        assert abs(float(a1) - meta[3]) < 1e-4 and abs(float(a5) - meta[4]) < 1e-4


def test_hidden_states_and_edges():
    g = load("hidden_micro")
    cfg, P = mg.oracle_cfg("micro"), params("micro")
    x, ci, ti = (torch.from_numpy(g[k]) for k in ("x", "ci", "ti"))
    with torch.no_grad():
        sos, ctx, tgt = orc.embed(P, cfg, x.reshape(2, -1), ci, ti)
        logits, hidden = orc.gpt_forward(P, cfg, sos, ctx, tgt, return_hidden=True)
    for i, (s, t) in enumerate(hidden):
        np.testing.assert_allclose(s.numpy(), g[f"sos{i}"], atol=TOL, rtol=0)
        np.testing.assert_allclose(t.numpy(), g[f"tgt{i}"], atol=TOL, rtol=0)
    np.testing.assert_allclose(logits[..., g["cols"]].numpy(), g["logits_cols"], atol=TOL, rtol=0)
    e = load("edges_micro")
    x, idx = torch.from_numpy(e["x"]), torch.from_numpy(e["indices"])
    with torch.no_grad():
        l0 = orc.reconstruct_mask(P, cfg, x, idx[:, :0], idx)            # NC = 0
        l1 = orc.reconstruct_mask(P, cfg, x, idx[:, :-1], idx[:, -1:])   # NT = 1
    assert torch.isfinite(l0).all()
    np.testing.assert_allclose(l0[..., e["cols"]].numpy(), e["nc0_cols"], atol=TOL, rtol=0)
    np.testing.assert_allclose(torch.logsumexp(l0.double(), -1).numpy(), e["nc0_lse"], atol=TOL, rtol=0)
    np.testing.assert_allclose(l1[..., e["cols"]].numpy(), e["nt1_cols"], atol=TOL, rtol=0)


def test_divide_indices():
    g = load("divide_indices")
    idx = torch.from_numpy(g["indices"])
    for k, (sched, num) in enumerate(zip(g["case_sched"], g["case_num"])):
        t, T, start, training, budget, seq_len = num
        cfg = orc.OracleConfig(1, 1, 8, 16, 1, ["latent_enc"], shape=(4, 2, 2), schedule=str(sched), budget=int(budget))
        c, tg, sl = orc.divide_indices(idx, float(t), cfg, bool(training), window=(int(T), int(start)))
        assert sl == int(seq_len)
        assert c.shape == g[f"k{k}_ctx"].shape and (c.numpy() == g[f"k{k}_ctx"]).all(), (k, sched, num)
        assert tg.shape == g[f"k{k}_tgt"].shape and (tg.numpy() == g[f"k{k}_tgt"]).all(), (k, sched, num)


def test_sample_from_logits_and_next_mask():
    g = load("sampler_ops")
    logits = torch.from_numpy(g["logits"])
    for i, (temp, k, p) in enumerate(g["cases"]):
        noise = torch.from_numpy(cf.exp1_noise("noise", tuple(logits.shape), stream=int(g[f"s{i}_stream"])))
        ids, probs = orc.sample_from_logits(logits, float(temp), None if k < 0 else int(k), None if p < 0 else float(p), noise)
        assert (ids.numpy() == g[f"s{i}_ids"]).all(), i
        np.testing.assert_allclose(probs.numpy(), g[f"s{i}_probs"], atol=1e-7, rtol=1e-6)
    ci, ti, score = (torch.from_numpy(g[k]) for k in ("g_ci", "g_ti", "g_score"))
    for i, (strategy, (ctemp, nm)) in enumerate(zip(g["g_strategy"], g["g_cases"])):
        noise_fn, _, st = mg.oracle_noise_fns(int(g[f"g{i}_stream"]))
        rn = noise_fn("randn", score.shape) if strategy in ("random", "bootstrap") else None
        nc, nt = orc.generate_next_mask(ci, ti, score, int(nm), str(strategy), float(ctemp),
                                        lambda: noise_fn("mask", score.shape), rn)
        assert (nc.numpy() == g[f"g{i}_ctx"]).all() and (nt.numpy() == g[f"g{i}_tgt"]).all(), (i, strategy)


def test_sample_loops():
    g = load("sample_loops")
    P = params("micro")
    for i, (strategy, sched, run) in enumerate(zip(g["run_strategy"], g["run_schedule"], g["runs"])):
        n_steps, temp, k, p, ctemp = run
        cfg = mg.oracle_cfg("micro", schedule=str(sched))
        noise_fn, _, st = mg.oracle_noise_fns()
        x = torch.zeros(2, 2, 4, 4, dtype=torch.long)
        with torch.no_grad():
            xs, ci, ti = orc.sample(P, cfg, x, int(n_steps), float(temp), None if k < 0 else int(k),
                                    None if p < 0 else float(p), float(ctemp), noise_fn, strategy=str(strategy))
        assert st["k"] == int(g[f"r{i}_ndraws"]), (i, st["k"])
        assert (xs.numpy() == g[f"r{i}_x"]).all(), i
        assert (ci.numpy() == g[f"r{i}_ci"]).all() and (ti.numpy() == g[f"r{i}_ti"]).all(), i
    cfg = mg.oracle_cfg("micro", schedule="cosine")
    noise_fn, _, _ = mg.oracle_noise_fns()
    idx = torch.from_numpy(g["cont_idx"])
    with torch.no_grad():
        xs, ci, ti = orc.sample(P, cfg, torch.from_numpy(g["cont_x0"]), 4, 1.0, None, None, 3.0, noise_fn,
                                ci=idx[:, :10], ti=idx[:, 10:])
    assert (xs.numpy() == g["cont_x"]).all() and (ci.numpy() == g["cont_ci"]).all() and (ti.numpy() == g["cont_ti"]).all()
    cfg = mg.oracle_cfg("micro")
    for i, r in enumerate(g["dnr"]):
        opt = lambda v, f: None if v < 0 else f(v)
        noise_fn, perm_fn, st = mg.oracle_noise_fns()
        with torch.no_grad():
            xs = orc.draft_and_revise(P, cfg, torch.from_numpy(g[f"d{i}_x0"]), int(r[0]), float(r[1]), opt(r[2], int),
                                      opt(r[3], float), int(r[4]), float(r[5]), opt(r[6], int), opt(r[7], float),
                                      int(r[8]), bool(r[9]), perm_fn, noise_fn)
        assert st["k"] == int(g[f"d{i}_ndraws"])
        assert (xs.numpy() == g[f"d{i}_x"]).all(), i


def test_script_drivers():
    """bidirect_sample / extrapolate (sample_vqgan_transformer_videos.py:22-157)."""
    g = load("script_drivers")
    P, cfg = params("micro"), mg.oracle_cfg("micro", schedule="cosine")
    with torch.no_grad():
        noise_fn, _, st = mg.oracle_noise_fns()
        cm, score = orc.bidirect_sample(P, cfg, 2, 8, 8, 4, 1.0, None, None, 4, 3.0, noise_fn, bootstrap=3)
        assert st["k"] == int(g["bi_ndraws"]) and (cm.numpy() == g["bi_code_maps"]).all()
        np.testing.assert_allclose(score.numpy(), g["bi_score"], rtol=1e-5)
        noise_fn, _, st = mg.oracle_noise_fns()
        cm, score = orc.bidirect_sample(P, cfg, 2, 8, 8, 4, 0.9, 64, None, 3, 2.0, noise_fn)
        assert st["k"] == int(g["bi2_ndraws"]) and (cm.numpy() == g["bi2_code_maps"]).all()
        np.testing.assert_allclose(score.numpy(), g["bi2_score"], rtol=1e-5)
        noise_fn, _, st = mg.oracle_noise_fns()
        cm = orc.extrapolate(P, cfg, torch.from_numpy(g["ex_vq0"]), 16, 8, 4, 1.0, None, None, 3, 2.5, noise_fn)
        assert st["k"] == int(g["ex_ndraws"]) and cm.shape == (2, 4, 4, 4) and (cm.numpy() == g["ex_code_maps"]).all()
        # sliding-window continuation of bidirect_sample (the reference cannot score this case): shape + context carry-over
        noise_fn, _, _ = mg.oracle_noise_fns()
        cm, score = orc.bidirect_sample(P, cfg, 2, 16, 8, 4, 1.0, None, None, 3, 2.0, noise_fn)
        assert cm.shape == (2, 4, 4, 4) and torch.isfinite(score).all()


@pytest.mark.parametrize("name", ["micro", "micro_budget"])
def test_train_steps(name):
    g = load("train_" + name)
    cfg, P = mg.oracle_cfg(name), params(name)
    names = [str(n) for n in g["names"]]
    decay, emb, no_decay, pos = orc.decay_split(P)
    assert [len(decay), len(emb), len(no_decay), len(pos)] == list(g["group_sizes"])
    st = orc.TrainState(P, lr=float(g["lr"]), weight_decay=float(g["wd"]))
    probe = g["probe"]
    for s, t in enumerate(g["ts"]):
        r = orc.train_step(st, cfg, torch.from_numpy(g[f"s{s}_x"]), torch.from_numpy(g[f"s{s}_indices"]), float(t))
        meta = g[f"s{s}_meta"]
        assert abs(r["loss"] - meta[0]) < 2e-5 * abs(meta[0]), (s, r["loss"], meta[0])
        assert abs(r["acc1"] - meta[1]) < 1e-4 and abs(r["acc5"] - meta[2]) < 1e-4
        gn = np.array([float(r["grads"][n].double().norm()) for n in names])
        np.testing.assert_allclose(gn, g[f"s{s}_gradnorm"], rtol=2e-4, atol=1e-7)
        pn = np.array([float(st.P[n].detach().double().norm()) for n in names])
        np.testing.assert_allclose(pn, g[f"s{s}_pnorm"], rtol=1e-5)
        pp = np.stack([st.P[n].detach().reshape(-1)[probe % st.P[n].numel()].numpy() for n in names])
        np.testing.assert_allclose(pp, g[f"s{s}_pprobe"], atol=3e-5, rtol=1e-3)


def test_windowed_train_step_c1():
    """A train step whose mask sampler sliced a temporal window (seq_len = 64 of N = 128: mask_sampler.py:83-99) on the C1
    config, against the reference (tests/golden/train_window_c1.npz; the reference's two numpy draws forced to T = 1, start 1)."""
    g = load("train_window_c1")
    cfg, P = mg.oracle_cfg("c1"), params("c1")
    names = [str(n) for n in g["names"]]
    st = orc.TrainState(P, lr=1e-3, weight_decay=0.05)
    assert list(g["choice_a"]) == [1, 2] and list(g["randint"]) == [0, 2]
    r = orc.train_step(st, cfg, torch.from_numpy(g["x"]), torch.from_numpy(g["indices"]), float(g["t"]), window=(1, 1))
    meta = g["meta"]
    assert abs(r["loss"] - meta[0]) < 2e-5 * abs(meta[0]) and r["n_targets"] == 3 * 39      # ceil(0.6 * 64) targets per sample
    assert abs(r["acc1"] - meta[1]) < 1e-4 and abs(r["acc5"] - meta[2]) < 1e-4
    gn = np.array([float(r["grads"][n].double().norm()) for n in names])
    np.testing.assert_allclose(gn, g["gradnorm"], rtol=2e-4, atol=1e-7)
    probe = g["probe"]
    gp = np.stack([r["grads"][n].reshape(-1)[probe % r["grads"][n].numel()].numpy() for n in names])
    np.testing.assert_allclose(gp, g["gprobe"], atol=2e-6, rtol=2e-3)
    pn = np.array([float(st.P[n].detach().double().norm()) for n in names])
    np.testing.assert_allclose(pn, g["pnorm"], rtol=1e-5)


def test_flop_model_matches_survey():
    cfg = orc.OracleConfig(24, 16, 1024, 1024, 256,
                           ["latent_enc", "latent_self"] * 6 + ["latent_enc"] + ["latent_dec", "lt2l"] * 5 + ["latent_dec"])
    assert abs(orc.forward_flops_per_sample(cfg, 512, 512) / 1e9 - 234.881) < 1e-3


@pytest.mark.parametrize("name", ["vq_micro", "vq_c5"])
def test_vqgan_oracle_matches_reference_golden(name):
    """oracle/vqgan_oracle.py (encode / decode of the 3D-VQGAN first stage) against vectors produced by the real reference
    `mebt.vqgan.VQGAN` (tests/golden/make_golden.py:gen_vqgan): token ids identical, pre-quantisation z and decoded video to
    fp32 rounding."""
    from oracle import vqgan_oracle as vq
    from tests.golden import make_golden as mg
    g = np.load(os.path.join(G, name + ".npz"))
    cfg = mg.vqgan_cfg(name)
    P = vq.closed_form_params(cfg)
    x = mg.vqgan_video(name)
    assert tuple(g["video_shape"]) == tuple(x.shape)
    with torch.no_grad():
        ids, z, d = vq.encode(P, cfg, x, return_all=True)
        flat = z.permute(0, 2, 3, 4, 1).reshape(-1, z.shape[1])
        rec = vq.decode(P, cfg, torch.from_numpy(g["dec_ids"]))
    np.testing.assert_allclose(flat[g["z_rows"]].numpy(), g["z_vals"], rtol=1e-4, atol=2e-5)
    best2 = torch.topk(d, 2, dim=1, largest=False).values.numpy()
    np.testing.assert_allclose(best2, g["best2"], rtol=1e-5, atol=1e-3)
    mism = (ids.numpy() != g["ids"])
    gap = (g["best2"][:, 1] - g["best2"][:, 0]).reshape(ids.shape)
    assert (gap[mism] < 1e-3).all() and mism.sum() <= 2                      # same code everywhere but at fp ties
    assert ids.dtype == torch.int64 and tuple(ids.shape) == tuple(g["ids"].shape)
    np.testing.assert_allclose(rec.reshape(-1)[g["rec_idx"]].numpy(), g["rec_vals"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(rec.mean(dim=(0, 1, 3, 4)).numpy(), g["rec_mean"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose((rec ** 2).mean(dim=(0, 1, 3, 4)).numpy(), g["rec_sq"], rtol=1e-4)
    assert tuple(rec.shape) == tuple(g["video_shape"][:1]) + (3,) + tuple(g["video_shape"][2:])


def test_inverse_cdf_twin_is_a_sample_of_the_reference_distribution():
    """oracle.sample_inverse_cdf - the CPU twin of the product's production draw (one uniform per row, inverse CDF in the kernel's
    element order) - against the reference's draw (arg-max p / q, q ~ Exp(1), mebt/transformer.py:826-841, restated and golden-pinned
    as oracle.sample_from_logits): both are ONE sample of the categorical distribution p left by temperature / top-k / softmax.
    Checked on 2048 rows of identical logits: both empirical distributions pass a chi-square test against p; the twin never takes an
    element the top-k filter removed, takes the only element of a one-hot row, and is a pure function of (seed, row)."""
    V, R = 16384, 2048
    row = torch.full((V,), -40.0)
    sup = torch.tensor([3, 511, 512, 1000, 4097, 8191, 8192, 12000, 16000, 16383, 700, 701])
    row[sup] = torch.tensor([2.0, 1.5, 1.0, 0.5, 0.0, -0.5, 1.2, 0.3, -1.0, 0.8, 0.1, 1.7])
    logits = row.repeat(R, 1)
    p = torch.softmax(row.double(), 0)
    ids, probs, margin = orc.sample_inverse_cdf(logits, 1.0, None, 20240607)
    counts = torch.bincount(ids, minlength=V).double()
    assert counts[sup].sum() == R
    chi2 = float((((counts[sup] - R * p[sup]) ** 2) / (R * p[sup])).sum())
    assert chi2 < 40.0, chi2                                                 # 11 degrees of freedom: P(chi2 > 40) = 4e-5
    noise = torch.empty(R, V).exponential_(generator=torch.Generator().manual_seed(5))
    rid, _ = orc.sample_from_logits(logits, 1.0, None, None, noise)
    rc = torch.bincount(rid, minlength=V).double()
    chi2r = float((((rc[sup] - R * p[sup]) ** 2) / (R * p[sup])).sum())
    assert chi2r < 40.0, chi2r
    ids2, _, _ = orc.sample_inverse_cdf(logits, 1.0, None, 20240607)
    assert torch.equal(ids, ids2)
    ids3, _, _ = orc.sample_inverse_cdf(logits, 1.0, None, 20240608)
    assert (ids3 != ids).float().mean() > 0.3
    # top-k 4: only the four largest logits (2.0, 1.7, 1.5, 1.2) are ever drawn, with their renormalised probabilities
    idk, pk, _ = orc.sample_inverse_cdf(logits[:1024], 1.0, 4, 77)
    keep = {3, 701, 511, 8192}
    assert set(idk.tolist()) <= keep and abs(float(pk[0].sum()) - 1.0) < 1e-5 and int((pk[0] > 0).sum()) == 4
    one = torch.full((2, V), -float("inf"))
    one[0, 777] = 0.3
    one[1, 16383] = -5.0
    ido, _, _ = orc.sample_inverse_cdf(one, 0.7, None, 1)
    assert ido.tolist() == [777, 16383]
    assert torch.equal(orc.inverse_cdf_order(V).sort().values, torch.arange(V))
