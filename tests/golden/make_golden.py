#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the *real* reference.

Runs ONLY in the build container (needs /root/reference, read-only).  The reference's Python never
travels to the GPU box: what is committed is data — inputs and (reduced) expected outputs — plus
this script.  Recipe: the reference package cannot be imported as-is (h5py,
pytorch_lightning, omegaconf, torchvision, imageio, skvideo are absent), so five small stand-ins
are registered *for third-party modules only* before importing `mebt.transformer`:

  1. empty namespace packages `mebt`, `mebt.modules` whose __path__ points at the reference dirs,
     so their __init__.py (which pulls h5py/torchvision/lpips) never runs;
  2. a fake `pytorch_lightning` exposing `LightningModule(nn.Module)`;
  3. empty `imageio`, `skvideo`, `skvideo.io` (import-time only, mebt/utils.py:3,8);
  4. a fake `mebt.download.load_vqgan` (imported unconditionally at transformer.py:181, never
     called under vtokens=True);
  5. attribute+item config objects (OmegaConf is absent).

Randomness is made machine-independent by replacing the three torch RNG entry points the path
uses (`Tensor.exponential_`, `torch.randn_like`, `torch.randperm`) and `random.random` with the
closed-form streams of oracle/closed_form.py, numbered in call order.

Usage:  python tests/golden/make_golden.py          (rewrites tests/golden/*.npz)
"""
import os
import sys
import types
import random

sys.dont_write_bytecode = True            # never write __pycache__ into /root/reference
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn as nn

from oracle import closed_form as cf
from oracle import mebt_oracle as orc


# ------------------------------------------------------------------------------------------------
# stubs
# ------------------------------------------------------------------------------------------------
def install_stubs():
    for name, path in (("mebt", [f"{REF}/mebt"]), ("mebt.modules", [f"{REF}/mebt/modules"])):
        m = types.ModuleType(name)
        m.__path__ = path
        sys.modules[name] = m

    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.global_step = 0
            self.current_epoch = 0
            self._logged = {}

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, name, value, **k):
            self._logged[name] = value

        @property
        def device(self):
            return next(self.parameters()).device

    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl
    for name in ("imageio", "skvideo", "skvideo.io"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["skvideo"].io = sys.modules["skvideo.io"]
    dl = types.ModuleType("mebt.download")
    dl.load_vqgan = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("vtokens only"))
    sys.modules["mebt.download"] = dl
    sys.path.insert(0, REF)               # top-level `utils.instantiate_from_config`


class Cfg(dict):
    """attr + item access, `in`, `.get` — what the reference needs from OmegaConf."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class ClosedFormRNG:
    """Numbered closed-form streams standing in for torch's global RNG."""
    def __init__(self):
        self.k = 0
        self.log = []

    def _next(self, kind, shape):
        k = self.k
        self.k += 1
        self.log.append((kind, tuple(shape)))
        return k

    def exponential_(self, tensor):
        k = self._next("exp", tensor.shape)
        tensor.copy_(torch.from_numpy(cf.exp1_noise("noise", tuple(tensor.shape), stream=k)))
        return tensor

    def randn_like(self, t):
        k = self._next("randn", t.shape)
        return torch.from_numpy(cf.pseudo_normal("noise", tuple(t.shape), std=1.0, stream=k))

    def randperm(self, n):
        k = self._next("perm", (n,))
        return torch.from_numpy(cf.permutation("noise", n, stream=k))


_REAL_RNG = (torch.Tensor.exponential_, torch.randn_like, torch.randperm)


def unpatch_rng():
    """torch's own generators again (the data-contract fixture draws from the seeded global torch RNG, as the reference's
    dataset does: its values depend on torch.randperm being the real one)"""
    torch.Tensor.exponential_, torch.randn_like, torch.randperm = _REAL_RNG


def patch_rng(rng):
    torch.Tensor.exponential_ = lambda self, *a, **k: rng.exponential_(self)
    torch.randn_like = lambda t, *a, **k: rng.randn_like(t)
    torch.randperm = lambda n, *a, **k: rng.randperm(n)


def oracle_noise_fns(start=0):
    """The same numbered streams, for replaying through the oracle (used by the tests too)."""
    state = {"k": start}

    def nxt():
        k = state["k"]
        state["k"] += 1
        return k

    def noise_fn(tag, shape):
        if tag == "randn":
            return torch.from_numpy(cf.pseudo_normal("noise", tuple(shape), std=1.0, stream=nxt()))
        return torch.from_numpy(cf.exp1_noise("noise", tuple(shape), stream=nxt()))

    def perm_fn(tag, B, N):
        return torch.stack([torch.from_numpy(cf.permutation("noise", N, stream=nxt())) for _ in range(B)])

    return noise_fn, perm_fn, state


# ------------------------------------------------------------------------------------------------
# model construction
# ------------------------------------------------------------------------------------------------
CONFIGS = {
    # BASELINE.json configs[0] / SURVEY.md §8 "C1": one block of each routing mode
    "c1": dict(n_layer=4, n_head=4, n_embd=256, block_size=256, sos_emb=64,
               mode=["latent_enc", "latent_self", "latent_dec", "lt2l"], shape=[2, 8, 8], budget=128),
    # micro config for per-block hidden states (SURVEY.md §8c fixture design)
    "micro": dict(n_layer=6, n_head=2, n_embd=64, block_size=32, sos_emb=8,
                  mode=["latent_enc", "latent_self", "latent_enc", "latent_dec", "lt2l", "latent_dec"],
                  shape=[2, 4, 4], budget=32),
    # micro config with a binding target budget and label smoothing
    "micro_budget": dict(n_layer=4, n_head=2, n_embd=64, block_size=32, sos_emb=8,
                         mode=["latent_enc", "latent_self", "latent_dec", "lt2l"],
                         shape=[2, 4, 4], budget=8, label_smoothing=0.1),
    # mode list shorter than n_layer -> padded with full-attention 'maskgit' blocks (gpt.py:208-209)
    "micro_maskgit": dict(n_layer=3, n_head=2, n_embd=64, block_size=32, sos_emb=8,
                          mode=["latent_enc", "latent_dec"], shape=[2, 4, 4], budget=32),
}


def oracle_cfg(name, schedule="linear"):
    c = CONFIGS[name]
    return orc.OracleConfig(c["n_layer"], c["n_head"], c["n_embd"], c["block_size"], c["sos_emb"],
                            c["mode"], shape=c["shape"], schedule=schedule, budget=c["budget"],
                            avg_loss=1.0, label_smoothing=c.get("label_smoothing", 0.0))


def build_reference(name, schedule="linear", **overrides):
    """`overrides`: extra / replaced keys of the transformer config node (beta_params, beta_iter, t_prior ...)"""
    from mebt.transformer import Net2NetTransformer
    c = CONFIGS[name]
    tcfg = Cfg(unconditional=True, vocab_size=16384, first_stage_vocab_size=16384,
               block_size=c["block_size"], n_layer=c["n_layer"], n_head=c["n_head"],
               n_embd=c["n_embd"], n_unmasked=0, embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0,
               sample_every_n_latent_frames=0, first_stage_key="video", cond_stage_key="label",
               vtokens=True, vtokens_pos=False, vis_epoch=100, sos_emb=c["sos_emb"], avg_loss=True,
               mode=list(c["mode"]), class_cond_dim=None)
    if "label_smoothing" in c:
        tcfg["label_smoothing"] = c["label_smoothing"]
    tcfg.update(overrides)
    mcfg = Cfg(target="mebt.mask_sampler.MaskGen",
               params=Cfg(iid=False, schedule=schedule, max_token=c["block_size"], method="mlm",
                          shape=c["shape"], t_range=[0.0, 1.0], budget=c["budget"]))
    model = Net2NetTransformer(tcfg, Cfg(params=Cfg(ckpt_path=None)), mcfg, cond_stage_key="label")
    ocfg = oracle_cfg(name, schedule)
    sd = {k: torch.from_numpy(v) for k, v in cf.state_dict_numpy(orc.param_shapes(ocfg)).items()}
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    return model, ocfg


def inputs(name, B, tag):
    c = CONFIGS[name]
    N = int(np.prod(c["shape"]))
    x = torch.from_numpy(cf.randint(f"x/{name}/{tag}", (B, *c["shape"]), 16384))
    idx = torch.stack([torch.from_numpy(cf.permutation(f"perm/{name}/{tag}", N, stream=b)) for b in range(B)])
    return x, idx


def digest(logits, cols):
    """Reduced description of a [B,NT,V] logits tensor (SURVEY.md §8c 'reduced outputs')."""
    lg = logits.detach().to(torch.float64)
    top_v, top_i = logits.detach().topk(5, dim=-1)
    return dict(lse=torch.logsumexp(lg, -1).numpy(), argmax=logits.argmax(-1).numpy(),
                top5_ids=top_i.numpy(), top5_vals=top_v.numpy(),
                cols=logits.detach()[..., cols].numpy(), mean=lg.mean(-1).numpy(),
                sqsum=(lg * lg).sum(-1).numpy())


COLS = cf.permutation("digest-cols", 16384)[:64].copy()


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(os.environ.get("MEBT_GOLDEN_OUT", HERE), name + ".npz")      # MEBT_GOLDEN_OUT: regenerate elsewhere (tests)
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------------
# fixtures
# ------------------------------------------------------------------------------------------------
def gen_forward():
    """Net2NetTransformer.forward + shared_step loss/acc (transformer.py:216-286, 717-732)."""
    for name, B in (("c1", 2), ("micro", 3), ("micro_budget", 3), ("micro_maskgit", 2)):
        model, _ = build_reference(name)
        x, idx = inputs(name, B, "fwd")
        out = {"x": x, "indices": idx, "cols": COLS}
        cases = [("train", 0.5), ("train", 0.13), ("eval", 0.71), ("train", 0.999), ("train", 0.0)]
        out["case_mode"] = np.array([c[0] for c in cases])
        out["case_t"] = np.array([c[1] for c in cases])
        for ci, (mode, t) in enumerate(cases):
            model.train(mode == "train")
            with torch.no_grad():
                logits, z_t, ntw, seq_len = model(x, None, t=t, indices=idx)
                # shared_step body (:723-731) on the same logits
                from mebt.utils import accuracy
                ratio = ntw / float(seq_len)
                ls = model.label_smoothing
                loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.size(-1)), z_t.reshape(-1),
                                                         reduction="sum", label_smoothing=ls)
                loss = loss / (B * seq_len * ratio ** model.config.avg_loss)
                a1, a5 = accuracy(logits.reshape(-1, logits.shape[-1]), z_t.reshape(-1), topk=(1, 5))
            d = digest(logits, COLS)
            for k, v in d.items():
                out[f"c{ci}_{k}"] = v
            out[f"c{ci}_z_targets"] = z_t
            out[f"c{ci}_meta"] = np.array([ntw, seq_len, float(loss), float(a1), float(a5)], dtype=np.float64)
        save(f"forward_{name}", **out)


def gen_hidden():
    """Per-block hidden states of the micro config (gpt.py:243-248)."""
    model, _ = build_reference("micro")
    model.eval()
    x, idx = inputs("micro", 2, "hidden")
    ci, ti = idx[:, :11], idx[:, 11:]
    hs = []
    hooks = [blk.register_forward_hook(lambda m, i, o: hs.append((o[0].detach().clone(), o[2].detach().clone())))
             for blk in model.transformer.blocks]
    with torch.no_grad():
        logits, _ = model.reconstruct_mask(x, ci, ti)
    for h in hooks:
        h.remove()
    out = {"x": x, "ci": ci, "ti": ti, "logits_cols": logits[..., COLS], "cols": COLS,
           "lse": torch.logsumexp(logits.double(), -1)}
    for i, (s, t) in enumerate(hs):
        out[f"sos{i}"] = s
        out[f"tgt{i}"] = t
    save("hidden_micro", **out)
    # NC = 0 edge (first sampling step) and NT = 1
    with torch.no_grad():
        l0, _ = model.reconstruct_mask(x, idx[:, :0], idx)
        l1, _ = model.reconstruct_mask(x, idx[:, :-1], idx[:, -1:])
    save("edges_micro", x=x, indices=idx, cols=COLS, nc0_cols=l0[..., COLS],
         nc0_lse=torch.logsumexp(l0.double(), -1), nt1_cols=l1[..., COLS],
         nt1_lse=torch.logsumexp(l1.double(), -1))


def gen_divide():
    """MaskGen.divide_indices incl. the sliced-window (video-length curriculum) branch
    (mask_sampler.py:75-115).  The two numpy draws are forced by patching np.random."""
    from mebt.mask_sampler import MaskGen
    out = {}
    shape = (4, 2, 2)
    N = 16
    idx = torch.stack([torch.from_numpy(cf.permutation("perm/divide", N, stream=b)) for b in range(3)])
    out["indices"] = idx
    cases = []
    k = 0
    for sched in ("linear", "cosine", "quadratic", "sqrt", "square", "cube", "cosine_plus", "convex"):
        for t in (0.0, 0.3, 0.77):
            for (T, start, training, budget) in ((4, 0, True, 16), (2, 1, True, 16), (1, 3, True, 16),
                                                 (4, 0, False, 5), (3, 0, True, 5)):
                mg = MaskGen(schedule=sched, shape=shape, budget=budget)
                mg.train(training)
                orig_choice, orig_randint = np.random.choice, np.random.randint
                np.random.choice = lambda a, p=None, _T=T: _T
                np.random.randint = lambda lo, hi=None, _s=start: _s
                try:
                    c, tg, sl = mg.divide_indices(idx, torch.tensor(t), np.arange(4) + 1, np.ones(4))
                finally:
                    np.random.choice, np.random.randint = orig_choice, orig_randint
                out[f"k{k}_ctx"], out[f"k{k}_tgt"] = c, tg
                cases.append((sched, t, T, start, int(training), budget, int(sl)))
                k += 1
    out["case_sched"] = np.array([c[0] for c in cases])
    out["case_num"] = np.array([c[1:] for c in cases], dtype=np.float64)
    save("divide_indices", **out)


def gen_sampler_ops():
    """sample_from_logits / top-k / top-p (transformer.py:826-910) and generate_next_mask
    (mask_sampler.py:178-246) on small tensors."""
    import mebt.transformer as T
    from mebt.mask_sampler import MaskGen
    rng = ClosedFormRNG()
    patch_rng(rng)
    out = {}
    V = 512
    logits = torch.from_numpy(cf.pseudo_normal("sfl/logits", (2, 5, V), std=2.0))
    logits[0, 0, 7] = logits[0, 0, 9]                     # a tie inside top-k
    out["logits"] = logits
    cases = [(1.0, None, None), (0.7, 8, None), (1.0, None, 0.9), (0.3, 16, 0.5), (0.0, None, None), (2.0, 1, None)]
    out["cases"] = np.array([[c[0], -1 if c[1] is None else c[1], -1 if c[2] is None else c[2]] for c in cases])
    for i, (temp, k, p) in enumerate(cases):
        out[f"s{i}_stream"] = np.array(rng.k)
        ids, probs = T.sample_from_logits(logits, temp, k, p, return_probs=True)
        out[f"s{i}_ids"], out[f"s{i}_probs"] = ids, probs
    # generate_next_mask
    mg = MaskGen(schedule="cosine", shape=(2, 4, 4))
    idx = torch.stack([torch.from_numpy(cf.permutation("perm/gnm", 32, stream=b)) for b in range(3)])
    score = torch.from_numpy(cf.uniform01("gnm/score", (3, 20)).astype(np.float32))
    ci, ti = idx[:, :12], idx[:, 12:]
    gcases = [("maskgit", 4.5, 15), ("maskgit", 0.0, 10), ("maskgit", 2.0, 25), ("random", 4.5, 12),
              ("bootstrap", 1.0, 3), ("mlm", 1.0, 0)]
    out["g_ci"], out["g_ti"], out["g_score"] = ci, ti, score
    out["g_cases"] = np.array([[c[1], c[2]] for c in gcases])
    out["g_strategy"] = np.array([c[0] for c in gcases])
    for i, (strategy, ctemp, nm) in enumerate(gcases):
        out[f"g{i}_stream"] = np.array(rng.k)
        nc, nt = mg.generate_next_mask(ci, ti, score, 0.5, strategy=strategy, context_temperature=ctemp,
                                       n_masked_toks=torch.full((3,), float(nm)))
        out[f"g{i}_ctx"], out[f"g{i}_tgt"] = nc, nt
    save("sampler_ops", **out)


def gen_sample_loops():
    """sample() (transformer.py:353-447) and draft_and_revise() (:632-663) on the micro config."""
    out = {}
    runs = [("maskgit", 6, 1.0, None, None, 6.0, "cosine"), ("maskgit", 4, 0.8, 32, None, 2.0, "linear"),
            ("random", 5, 1.0, None, 0.95, 4.5, "cosine"), ("bootstrap", 3, 1.0, None, None, 1.0, "cosine")]
    out["runs"] = np.array([[r[1], r[2], -1 if r[3] is None else r[3], -1 if r[4] is None else r[4], r[5]] for r in runs])
    out["run_strategy"] = np.array([r[0] for r in runs])
    out["run_schedule"] = np.array([r[6] for r in runs])
    for i, (strategy, n_steps, temp, k, p, ctemp, sched) in enumerate(runs):
        model, _ = build_reference("micro", schedule=sched)
        model.eval()
        rng = ClosedFormRNG()
        patch_rng(rng)
        x = torch.zeros(2, 2, 4, 4, dtype=torch.long)
        with torch.no_grad():
            xs, ci, ti = model.sample(x, None, temp, k, p, n_steps, None, None, strategy=strategy,
                                      context_temperature=ctemp, skips=False)
        out[f"r{i}_x"], out[f"r{i}_ci"], out[f"r{i}_ti"] = xs, ci, ti
        out[f"r{i}_ndraws"] = np.array(rng.k)
    # sample() continuing from given context/target sets (sliding-window style call,
    # sample_vqgan_transformer_videos.py:65) with edit=False
    model, _ = build_reference("micro", schedule="cosine")
    model.eval()
    rng = ClosedFormRNG()
    patch_rng(rng)
    x0, idx = inputs("micro", 2, "cont")
    with torch.no_grad():
        xs, ci, ti = model.sample(x0, None, 1.0, None, None, 4, idx[:, :10], idx[:, 10:], context_temperature=3.0, skips=False)
    out["cont_x0"], out["cont_idx"], out["cont_x"], out["cont_ci"], out["cont_ti"] = x0, idx, xs, ci, ti
    # draft_and_revise
    dnr = [(4, 1.0, None, None, 2, 0.7, None, None, 2, False), (8, 0.0, None, None, 4, 0.3, 16, 0.9, 1, False),
           (4, 1.0, None, None, 8, 1.0, None, None, 2, True)]
    out["dnr"] = np.array([[-1 if v is None else float(v) for v in r] for r in dnr])
    for i, r in enumerate(dnr):
        model, _ = build_reference("micro")
        model.eval()
        rng = ClosedFormRNG()
        patch_rng(rng)
        x0, _ = inputs("micro", 2, f"dnr{i}")
        with torch.no_grad():
            xs = model.draft_and_revise(x0, None, *r)
        out[f"d{i}_x0"], out[f"d{i}_x"] = x0, xs
        out[f"d{i}_ndraws"] = np.array(rng.k)
    save("sample_loops", **out)


def gen_train():
    """Three optimiser steps of the reference training step (shared_step :717-732, backward,
    configure_optimizers :749-798, optimizer_step :665-681) on the micro configs."""
    for name in ("micro", "micro_budget"):
        model, ocfg = build_reference(name)
        model.train()
        model.learning_rate, model.weight_decay = 1e-3, 0.05
        opt = model.configure_optimizers()
        ts = [0.5, 0.21, 0.83]
        out = {"ts": np.array(ts), "lr": np.array(1e-3), "wd": np.array(0.05)}
        out["group_sizes"] = np.array([len(list(g["params"])) for g in opt.param_groups])
        out["group_wd"] = np.array([g["weight_decay"] for g in opt.param_groups])
        names = sorted(orc.param_shapes(ocfg).keys())
        probe = cf.permutation("train-probe", 4096)[:16]
        for s, t in enumerate(ts):
            x, idx = inputs(name, 3, f"train{s}")
            out[f"s{s}_x"], out[f"s{s}_indices"] = x, idx
            opt.zero_grad()
            acc1, acc5, loss, ratio = _shared_step_with_t(model, x, idx, t)
            loss.backward()
            sd = dict(model.named_parameters())
            out[f"s{s}_meta"] = np.array([float(loss), float(acc1), float(acc5)])
            out[f"s{s}_gradnorm"] = np.array([float(sd[n].grad.double().norm()) if sd[n].grad is not None else 0.0 for n in names])
            opt.step()
            out[f"s{s}_pnorm"] = np.array([float(sd[n].detach().double().norm()) for n in names])
            out[f"s{s}_pprobe"] = np.stack([sd[n].detach().reshape(-1)[probe % sd[n].numel()].numpy() for n in names])
        out["names"] = np.array(names)
        out["probe"] = probe
        save(f"train_{name}", **out)


def gen_script_drivers():
    """bidirect_sample / extrapolate of reference sample_vqgan_transformer_videos.py:22-157 on the
    micro config.  Only import-time names of the script are stubbed (matplotlib, omegaconf, the heavy
    `mebt` package attributes); the pixel decode (3D-VQGAN, outside this path) is a dummy."""
    import importlib.util
    for name in ("matplotlib", "matplotlib.pyplot", "omegaconf", "mebt.data", "pytorch_lightning.callbacks"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    sys.modules["omegaconf"].OmegaConf = object
    sys.modules["mebt.data"].preprocess = None
    sys.modules["pytorch_lightning.callbacks"].ModelCheckpoint = object
    sys.modules["pytorch_lightning"].callbacks = sys.modules["pytorch_lightning.callbacks"]
    import mebt
    import mebt.utils as mu
    from mebt.transformer import Net2NetTransformer
    for n in ("VideoData", "load_vqgan", "load_transformer"):
        setattr(mebt, n, None)
    mebt.Net2NetTransformer = Net2NetTransformer
    if not hasattr(mu, "save_video_grid"):
        mu.save_video_grid = None
    spec = importlib.util.spec_from_file_location("ref_sample_script", f"{REF}/sample_vqgan_transformer_videos.py")
    script = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(script)

    class DummyDecoder:                      # stands in for the VQGAN decode: [B,T,H,W] ids -> [B,3,4T,8H,8W] zeros
        def decode(self, code_map):
            b, t, h, w = code_map.shape
            return torch.zeros(b, 3, 4 * t, 8 * h, 8 * w)

    out = {}
    model, _ = build_reference("micro", schedule="cosine")
    model.eval()
    model.first_stage_model = DummyDecoder()
    rng = ClosedFormRNG()
    patch_rng(rng)
    log = script.bidirect_sample(model, 2, 8, 8, 4, temperature=1.0, top_k=None, top_p=None, vid_n_steps=4, vid_c_temp=3.0,
                                 ctemp_schedule='linear', strategy='maskgit', bootstrap=3)
    out["bi_code_maps"], out["bi_score"], out["bi_ndraws"] = log["code_maps"], log["score"], np.array(rng.k)
    rng = ClosedFormRNG()
    patch_rng(rng)
    log = script.bidirect_sample(model, 2, 8, 8, 4, temperature=0.9, top_k=64, top_p=None, vid_n_steps=3, vid_c_temp=2.0,
                                 ctemp_schedule='linear', strategy='maskgit', bootstrap=0)
    out["bi2_code_maps"], out["bi2_score"], out["bi2_ndraws"] = log["code_maps"], log["score"], np.array(rng.k)
    rng = ClosedFormRNG()
    patch_rng(rng)
    vq0, _ = inputs("micro", 2, "extrap")
    log = script.extrapolate(model, vq0, 16, 8, 4, temperature=1.0, top_k=None, top_p=None, vid_n_steps=3, vid_c_temp=2.5)
    out["ex_vq0"], out["ex_code_maps"], out["ex_ndraws"] = vq0, log["code_maps"], np.array(rng.k)
    save("script_drivers", **out)


def gen_data_contract():
    """Items of the reference's HDF5Dataset_vtokens (mebt/data.py:330-414) on a small synthetic token file.  h5py is not
    installed here: `h5py.File` is stubbed by an npz reader (a container stub, the dataset logic is the reference's)."""
    import importlib
    unpatch_rng()
    tmp = os.path.join(os.environ.get("MEBT_GOLDEN_OUT", HERE), "_tokens_tmp.npz")
    rs = np.random.RandomState(7)
    lens = [9, 3, 12, 5, 20, 4, 7]                       # frames per video; sequence_length 4 -> videos 1 and 5 are too short
    idx = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tokens = rs.randint(0, 16384, size=(int(idx[-1]), 6, 6)).astype(np.int64)
    test_lens = [6, 8]
    tidx = np.concatenate([[0], np.cumsum(test_lens)]).astype(np.int64)
    ttokens = rs.randint(0, 16384, size=(int(tidx[-1]), 6, 6)).astype(np.int64)
    np.savez(tmp, train_data=tokens, train_idx=idx, test_data=ttokens, test_idx=tidx)
    h5 = types.ModuleType("h5py")
    h5.File = lambda path, mode="r": dict(np.load(path))
    sys.modules["h5py"] = h5
    for name in ("torchvision", "torchvision.transforms", "torchvision.datasets", "torchvision.datasets.video_utils", "PIL", "PIL.Image", "av"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision.datasets.video_utils"].VideoClips = object
    if not hasattr(sys.modules["pytorch_lightning"], "LightningDataModule"):
        sys.modules["pytorch_lightning"].LightningDataModule = object
    if not getattr(sys.modules.get("mebt.data"), "__file__", None):     # gen_script_drivers' import-time stub: drop it, import the real one
        sys.modules.pop("mebt.data", None)
    try:
        data_mod = importlib.import_module("mebt.data")
    except Exception as e:                                 # heavy optional imports of the module: stub what is missing, once more
        missing = str(e)
        raise RuntimeError("could not import reference mebt.data: " + missing)
    out = {"train_data": tokens, "train_idx": idx, "test_data": ttokens, "test_idx": tidx}
    cases = [("full", dict(sequence_length=4, resolution=6, spatial_length=6, sample_every_n_frames=1, latent_shape=[4, 6, 6])),
             ("crop", dict(sequence_length=4, resolution=6, spatial_length=4, sample_every_n_frames=1, latent_shape=[4, 4, 4])),
             ("skip", dict(sequence_length=6, resolution=6, spatial_length=6, sample_every_n_frames=2, latent_shape=[3, 6, 6]))]
    for tag, kw in cases:
        for train in (True, False):
            ds = data_mod.HDF5Dataset_vtokens(tmp, train=train, **kw)
            torch.manual_seed(123)
            vids, boxes, perms = [], [], []
            for i in range(len(ds)):
                it = ds[i]
                vids.append(it["video"].numpy()); perms.append(it["indices"].numpy())
                boxes.append(np.asarray(it["cbox"]).reshape(-1) if not np.isscalar(it["cbox"]) else np.zeros(4, np.int64))
            pre = f"{tag}_{'train' if train else 'test'}"
            out[pre + "_video"], out[pre + "_cbox"], out[pre + "_indices"] = np.stack(vids), np.stack(boxes), np.stack(perms)
    os.remove(tmp)
    save("data_contract", **out)


# ------------------------------------------------------------------------------------------------
# 3D-VQGAN first stage: VQGAN.encode / VQGAN.decode of the real reference
# ------------------------------------------------------------------------------------------------
VQGAN_CONFIGS = {
    # small: every layer kind at toy size ([1,3,4,16,16] -> [2,4,4]); channel counts 16/32/64 also exercise the kernels'
    # non-MFMA path (Cin = 16)
    "vq_micro": dict(n_hiddens=16, downsample=(2, 4, 4), embedding_dim=64, n_codes=512, video=(2, 3, 4, 16, 16)),
    # BASELINE.json configs[4] geometry (TATS-style values recalled in SURVEY.md §8f: not in the reference repository, whose
    # argparse defaults are 240 / (4,4,4) / 2048): [1,3,16,128,128] -> [1,4,16,16], 16384 codes of dimension 256
    "vq_c5": dict(n_hiddens=32, downsample=(4, 8, 8), embedding_dim=256, n_codes=16384, video=(1, 3, 16, 128, 128)),
}


def vqgan_cfg(name):
    from oracle import vqgan_oracle as vq
    c = VQGAN_CONFIGS[name]
    return vq.VQGANConfig(c["n_hiddens"], c["downsample"], 3, c["embedding_dim"], c["n_codes"])


def vqgan_video(name):
    """closed-form video in [-0.5, 0.5] (the reference's data range, data.py preprocess) with smooth structure + noise"""
    B, C, T, H, W = VQGAN_CONFIGS[name]["video"]
    u = cf.uniform01(f"vqgan-video/{name}", (B, C, T, H, W)).astype(np.float32)
    t = np.linspace(0, 1, T, dtype=np.float32)[None, None, :, None, None]
    y = np.linspace(0, 1, H, dtype=np.float32)[None, None, None, :, None]
    xg = np.linspace(0, 1, W, dtype=np.float32)[None, None, None, None, :]
    c = np.arange(C, dtype=np.float32)[None, :, None, None, None]
    base = 0.3 * np.sin(6.0 * xg + 2.0 * c + 3.0 * t) * np.cos(5.0 * y - t)
    return torch.from_numpy(np.clip(base + 0.4 * (u - 0.5), -0.5, 0.5).astype(np.float32))


def build_reference_vqgan(name):
    """the real mebt.vqgan.VQGAN with closed-form weights; LPIPS (torchvision VGG) is the only stand-in"""
    import argparse
    from oracle import vqgan_oracle as vq
    mods = sys.modules["mebt.modules"]
    if not hasattr(mods, "Codebook"):
        import importlib
        mods.Codebook = importlib.import_module("mebt.modules.codebook").Codebook

        class LPIPS(nn.Module):            # perceptual loss network: training only, never touched by encode / decode
            def forward(self, a, b):
                raise RuntimeError("stub")
        mods.LPIPS = LPIPS
    from mebt.vqgan import VQGAN
    c = VQGAN_CONFIGS[name]
    args = argparse.Namespace(embedding_dim=c["embedding_dim"], n_codes=c["n_codes"], n_hiddens=c["n_hiddens"], downsample=c["downsample"],
                              image_channels=3, norm_type="group", padding_type="replicate", no_random_restart=False, restart_thres=1.0,
                              gan_feat_weight=0.0, disc_channels=64, disc_layers=3, disc_loss_type="hinge", image_gan_weight=1.0,
                              video_gan_weight=1.0, perceptual_weight=0.0, l1_weight=4.0, sequence_length=c["video"][2],
                              sample_every_n_frames=1, resolution=c["video"][3], lr=3e-4, discriminator_iter_start=50000)
    model = VQGAN(args)
    cfg = vqgan_cfg(name)
    P = vq.closed_form_params(cfg)
    missing, unexpected = model.load_state_dict(P, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith(("image_discriminator", "video_discriminator", "perceptual_model", "codebook.N", "codebook.z_avg")) for k in missing), missing
    model.codebook._need_init = False      # transformer.py:186
    return model.eval(), cfg, P


def gen_vqgan():
    """encode: token ids (complete) + the pre-quantisation z at fixed positions + the best / second-best distance of every
    position (so that a consumer can tell an fp near-tie from a real mismatch); decode of fixed closed-form ids: the
    video at fixed voxels + per-frame mean / mean-square."""
    for name in VQGAN_CONFIGS:
        model, cfg, P = build_reference_vqgan(name)
        x = vqgan_video(name)
        with torch.no_grad():
            emb, ids = model.encode(x, include_embeddings=True)
            z = model.pre_vq_conv(model.encoder(x))
            flat = z.permute(0, 2, 3, 4, 1).reshape(-1, z.shape[1])
            d = (flat ** 2).sum(1, keepdim=True) - 2 * flat @ model.codebook.embeddings.t() + (model.codebook.embeddings.t() ** 2).sum(0, keepdim=True)
            best2 = torch.topk(d, 2, dim=1, largest=False).values
            assert torch.equal(d.argmin(1).view(ids.shape), ids)
            B, t_, h_, w_ = ids.shape
            dec_ids = torch.from_numpy(cf.randint(f"vqgan-ids/{name}", (B, t_, h_, w_), cfg.n_codes))
            rec = model.decode(dec_ids)
        zi = cf.permutation(f"vqgan-zpos/{name}", flat.shape[0])[:64].copy()
        vi = cf.permutation(f"vqgan-vox/{name}", rec.numel())[:4096].copy()
        save(name, video_shape=np.array(x.shape), ids=ids, z_rows=zi, z_vals=flat[zi], best2=best2, emb_check=emb[0, :8, 0, 0, 0],
             dec_ids=dec_ids, rec_idx=vi, rec_vals=rec.reshape(-1)[vi], rec_mean=rec.mean(dim=(0, 1, 3, 4)), rec_sq=(rec ** 2).mean(dim=(0, 1, 3, 4)))


def gen_edit_beta_window():
    """Branches of the hot path the first fixtures did not reach: `edit=True` in sample / draft_and_revise
    (transformer.py:373-376,399,658-660), the beta(t) schedule of the training `t` (:113-119,229-241), the video-length
    priors (:24-49) and a WINDOWED train step — divide_indices slicing a temporal window (mask_sampler.py:83-99), so that
    seq_len < N — on the C1 config.  Library RNG draws are forced AND recorded: the arguments the reference hands to
    Beta(alpha, beta), np.random.choice(vid_t, p=prior) and np.random.randint are part of the fixture."""
    out = {}
    # ---- edit=True: sample() continuing from index sets, schedule counted on the edited tokens only
    model, _ = build_reference("micro", schedule="cosine")
    model.eval()
    rng = ClosedFormRNG()
    patch_rng(rng)
    x0, idx = inputs("micro", 2, "edit")
    with torch.no_grad():
        xs, ci, ti = model.sample(x0, None, 1.0, None, None, 5, idx[:, :12], idx[:, 12:], context_temperature=3.0, skips=False, edit=True)
    out["edit_x0"], out["edit_idx"], out["edit_x"], out["edit_ci"], out["edit_ti"] = x0, idx, xs, ci, ti
    out["edit_ndraws"] = np.array(rng.k)
    # ---- edit=True: draft_and_revise with caller-given index sets for the draft, full revise afterwards
    model, _ = build_reference("micro")
    model.eval()
    rng = ClosedFormRNG()
    patch_rng(rng)
    x0, idx = inputs("micro", 2, "edit-dnr")
    with torch.no_grad():
        xs = model.draft_and_revise(x0, None, 4, 1.0, None, None, 4, 0.7, None, None, 2, False, False, idx[:, :16], idx[:, 16:], True)
    out["edit_dnr_x0"], out["edit_dnr_idx"], out["edit_dnr_x"] = x0, idx, xs
    out["edit_dnr_ndraws"] = np.array(rng.k)

    # ---- video-length priors at a few global steps (pure numpy, transformer.py:24-49)
    import mebt.transformer as rt
    steps = np.array([0, 1, 15000, 30000, 45000, 90000, 250000])
    lengths = np.arange(4) + 1
    for name in ("uniform", "gaussian2", "gaussian100000_2", "longest"):
        out["prior_" + name] = np.stack([getattr(rt, name)(lengths, int(s)) for s in steps])
    out["prior_steps"], out["prior_lengths"] = steps, lengths

    # ---- beta(t): what the reference hands to Beta() at given global steps, and a train-mode forward with the drawn t forced
    import torch.distributions.beta as tdb
    model, ocfg = build_reference("micro", beta_params=[3.0, 9.0], beta_iter=1000)
    model.train()
    calls = []
    forced = [0.35, 0.62, 0.18, 0.5, 0.77]
    real_beta = tdb.Beta

    class FakeBeta:
        def __init__(self, a, b):
            calls.append((float(a), float(b)))

        def sample(self):
            return torch.tensor(forced[len(calls) - 1])
    tdb.Beta = FakeBeta
    torch.distributions.beta.Beta = FakeBeta
    try:
        gsteps = [0, 250, 500, 1000, 1001]
        losses = []
        for s, gs in enumerate(gsteps):
            model.global_step = gs
            x, idx = inputs("micro", 2, f"beta{s}")
            out[f"beta{s}_x"], out[f"beta{s}_indices"] = x, idx
            with torch.no_grad():
                acc1, acc5, loss, ratio = model.shared_step({"video": x, "label": x, "indices": idx}, 0)
            losses.append([float(loss), float(acc1), float(acc5), float(ratio)])
    finally:
        tdb.Beta = real_beta
        torch.distributions.beta.Beta = real_beta
    out["beta_params"], out["beta_iter"] = np.array([3.0, 9.0]), np.array(1000.0)
    out["beta_gsteps"], out["beta_calls"], out["beta_forced_t"], out["beta_meta"] = np.array(gsteps), np.array(calls), np.array(forced), np.array(losses)
    save("edit_beta_priors", **out)

    # ---- windowed train step on C1 (t_prior = gaussian2): one optimizer step with T = 1 of 2 latent frames, start frame 1
    out = {}
    model, ocfg = build_reference("c1", t_prior="gaussian2")
    model.train()
    model.learning_rate, model.weight_decay = 1e-3, 0.05
    model.global_step = 12345
    opt = model.configure_optimizers()
    rec = {}
    real_choice, real_randint = np.random.choice, np.random.randint

    def fake_choice(a, p=None, **kw):
        rec["choice_a"], rec["choice_p"] = np.asarray(a).copy(), np.asarray(p).copy()
        return 1
    def fake_randint(lo, hi=None, **kw):
        rec["randint"] = (lo, hi)
        return 1
    np.random.choice, np.random.randint = fake_choice, fake_randint
    try:
        x, idx = inputs("c1", 3, "window")
        opt.zero_grad()
        acc1, acc5, loss, ratio = _shared_step_with_t(model, x, idx, 0.4)
        loss.backward()
    finally:
        np.random.choice, np.random.randint = real_choice, real_randint
    names = sorted(orc.param_shapes(ocfg).keys())
    sd = dict(model.named_parameters())
    probe = cf.permutation("train-probe", 4096)[:16]
    out["x"], out["indices"], out["t"], out["global_step"] = x, idx, np.array(0.4), np.array(12345)
    out["choice_a"], out["choice_p"], out["randint"] = rec["choice_a"], rec["choice_p"], np.array(rec["randint"])
    out["meta"] = np.array([float(loss), float(acc1), float(acc5), float(ratio)])
    out["gradnorm"] = np.array([float(sd[n].grad.double().norm()) if sd[n].grad is not None else 0.0 for n in names])
    out["gprobe"] = np.stack([(sd[n].grad if sd[n].grad is not None else torch.zeros_like(sd[n])).reshape(-1)[probe % sd[n].numel()].numpy() for n in names])
    opt.step()
    out["pnorm"] = np.array([float(sd[n].detach().double().norm()) for n in names])
    out["pprobe"] = np.stack([sd[n].detach().reshape(-1)[probe % sd[n].numel()].numpy() for n in names])
    out["names"], out["probe"] = np.array(names), probe
    save("train_window_c1", **out)


def gen_utils():
    """mebt/utils.py helpers on the path: shift_dim (:30-53; used by the VQGAN and the sampling scripts) and accuracy (:80-94)."""
    import mebt.utils as mu
    out = {}
    x = torch.from_numpy(cf.pseudo_normal("utils/x", (2, 3, 4, 5, 6), std=1.0))
    cases = [(1, -1), (-1, 1), (0, 2), (3, 1), (2, 2)]
    out["x"], out["shift_cases"] = x, np.array(cases)
    for i, (a, b) in enumerate(cases):
        out[f"shift{i}"] = mu.shift_dim(x, a, b).contiguous()
    logits = torch.from_numpy(cf.pseudo_normal("utils/logits", (64, 50), std=2.0))
    target = torch.from_numpy(cf.randint("utils/target", (64,), 50))
    out["acc_logits"], out["acc_target"] = logits, target
    a1, a5 = mu.accuracy(logits, target, topk=(1, 5))
    out["acc"] = np.array([float(a1), float(a5)])
    save("utils", **out)


def _shared_step_with_t(model, x, idx, t):
    """shared_step (:717-732) with the python RNG draw `t` (:228) forced."""
    orig = random.random
    random.random = lambda: t
    try:
        return model.shared_step({"video": x, "label": x, "indices": idx}, 0)
    finally:
        random.random = orig


GENERATORS = ["gen_forward", "gen_hidden", "gen_divide", "gen_sampler_ops", "gen_sample_loops", "gen_train", "gen_edit_beta_window", "gen_utils",
              "gen_script_drivers", "gen_data_contract", "gen_vqgan"]


def main(only=None):
    """python tests/golden/make_golden.py [gen_name ...]  — all generators run in ONE process, in this order (the later ones
    register import-time stand-ins for optional third-party modules that the earlier ones must not see)."""
    install_stubs()
    torch.manual_seed(0)
    for g in GENERATORS:
        if only and g not in only:
            continue
        globals()[g]()


if __name__ == "__main__":
    main(sys.argv[1:] or None)
