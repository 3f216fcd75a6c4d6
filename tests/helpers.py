"""Shared helpers of the GPU parity tests (test infrastructure; may import oracle/)."""
import os
import numpy as np
import torch

from oracle import mebt_oracle as orc

G = os.path.join(os.path.dirname(__file__), "golden")
_params = {}


def load_golden(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def params_for(name):
    from tests.golden import make_golden as mg
    if name not in _params:
        _params[name] = orc.closed_form_params(mg.oracle_cfg(name))
    return _params[name]


def build_native(cfg, dtype, P=None, device="cuda"):
    """NativeModel (HIP engine) holding the closed-form weights of `cfg`."""
    from mebt_amd.engine import NativeModel
    nm = NativeModel(cfg.n_layer, cfg.n_head, cfg.n_embd, cfg.vocab_size, cfg.sos_emb, cfg.block_size, cfg.mode,
                     dtype=dtype, label_smoothing=cfg.label_smoothing)
    nm.allocate(device)
    if P is None:
        P = orc.closed_form_params(cfg)
    views = nm.views(orc.param_shapes(cfg))
    with torch.no_grad():
        for k, v in P.items():
            views[k].copy_(v.detach())
    nm.sync_lowp(force=True)
    return nm


def product_config(name, schedule="linear", vtokens=True, **overrides):
    """Config objects for the shipped Net2NetTransformer equal to make_golden.build_reference's (`overrides`: extra keys of
    the transformer node, as there)."""
    from mebt_amd.config import AttrDict
    from tests.golden import make_golden as mg
    c = mg.CONFIGS[name]
    tcfg = AttrDict(unconditional=True, vocab_size=16384, first_stage_vocab_size=16384, block_size=c["block_size"],
                    n_layer=c["n_layer"], n_head=c["n_head"], n_embd=c["n_embd"], n_unmasked=0, embd_pdrop=0.0,
                    resid_pdrop=0.0, attn_pdrop=0.0, sample_every_n_latent_frames=0, first_stage_key="video",
                    cond_stage_key="label", vtokens=vtokens, vtokens_pos=False, vis_epoch=100, sos_emb=c["sos_emb"],
                    avg_loss=True, mode=list(c["mode"]), class_cond_dim=None)
    if "label_smoothing" in c:
        tcfg["label_smoothing"] = c["label_smoothing"]
    tcfg.update(overrides)
    mcfg = AttrDict(target="mebt.mask_sampler.MaskGen",
                    params=AttrDict(iid=False, schedule=schedule, max_token=c["block_size"], method="mlm",
                                    shape=c["shape"], t_range=[0.0, 1.0], budget=c["budget"]))
    return tcfg, AttrDict(params=AttrDict(ckpt_path=None)), mcfg


def build_product(name, dtype, schedule="linear", device="cuda", **overrides):
    """The shipped module (mebt.transformer.Net2NetTransformer) with the closed-form weights."""
    from mebt.transformer import Net2NetTransformer
    from oracle import closed_form as cf
    from tests.golden import make_golden as mg
    tcfg, vcfg, mcfg = product_config(name, schedule, **overrides)
    model = Net2NetTransformer(tcfg, vcfg, mcfg, cond_stage_key="label")
    model.compute_dtype = dtype
    sd = {k: torch.from_numpy(v) for k, v in cf.state_dict_numpy(orc.param_shapes(mg.oracle_cfg(name, schedule))).items()}
    model.load_state_dict(sd, strict=True)
    return model.to(device)


def closed_form_hook():
    """noise hook replaying the numbered closed-form streams of make_golden.ClosedFormRNG"""
    from oracle import closed_form as cf
    state = {"k": 0}

    def hook(kind, shape):
        k = state["k"]
        state["k"] += 1
        if kind == "randn":
            return torch.from_numpy(cf.pseudo_normal("noise", tuple(shape), std=1.0, stream=k))
        if kind == "perm":
            return torch.from_numpy(cf.permutation("noise", int(shape[0]), stream=k))
        return torch.from_numpy(cf.exp1_noise("noise", tuple(shape), stream=k))

    return hook, state


def record_measured(name, value, gate, note=""):
    """Append a measured error figure and its gate to gpurun_out/parity_measured.txt (copied to profiles/ after a GPU run), so
    that drift of the bf16 bounds is visible from round to round.  Never fails a test."""
    try:
        root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        d = os.path.join(root, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_measured.txt"), "a") as f:
            f.write(f"{name}\tmeasured {float(value):.4e}\tgate {float(gate):.4e}\t{note}\n")
    except OSError:
        pass


def assert_same_trajectory(a, b, what, lr=1e-3):
    """Two runs of the same training steps that differ only in the order of atomically accumulated sums (embedding rows, bias
    column sums): last-bit differences of a gradient become up to ~1e-6 on single parameters whose gradient sits near AdamW's
    epsilon, so single elements are bounded at a twentieth of one step and the mean at rounding level — a run that dropped a
    moment, a step count, a seed or a schedule position moves every element by ~lr."""
    import torch
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    d = (a - b).abs()
    scale = 1.0 + a.abs().max().item()
    assert d.max().item() <= 0.05 * lr * scale, (what, d.max().item())
    assert d.mean().item() <= 1e-6 * scale, (what, d.mean().item())
