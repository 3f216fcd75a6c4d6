"""Named hyper-parameter presets equal to the reference's shipped YAMLs (values cited, files not
copied): the reference configs load unchanged through mebt_amd.config.load_config; these presets
exist so that bench.py / tests need no file from the reference tree."""
from .config import AttrDict

_SKY_MODES = (["latent_enc", "latent_self"] * 6 + ["latent_enc"] + ["latent_dec", "lt2l"] * 5 + ["latent_dec"])


def sky_16f(vtokens=True, dropout=0.1):
    """configs/stl/mebt_16f.yaml:4-57,82 — 24L / d=1024 / 16 heads / 256 latents / 1024 tokens"""
    return AttrDict(model=AttrDict(
        target="mebt.transformer.Net2NetTransformer",
        params=AttrDict(unconditional=True, vocab_size=16384, first_stage_vocab_size=16384, block_size=1024,
                        n_layer=24, n_head=16, n_embd=1024, n_unmasked=0, embd_pdrop=dropout, resid_pdrop=dropout,
                        attn_pdrop=dropout, sample_every_n_latent_frames=0, first_stage_key="video",
                        cond_stage_key="label", vtokens=vtokens, vtokens_pos=False, vis_epoch=100, sos_emb=256,
                        avg_loss=True, mode=list(_SKY_MODES), class_cond_dim=None),
        mask=AttrDict(target="mebt.mask_sampler.MaskGen",
                      params=AttrDict(iid=False, schedule="linear", max_token=1024, method="mlm", shape=[4, 16, 16],
                                      t_range=[0.0, 1.0], budget=1024)),
        vqvae=AttrDict(params=AttrDict(ckpt_path=None, ignore_keys=["loss"]))),
        data=AttrDict(batch_size=6, sequence_length=16, resolution=128),
        exp=AttrDict(exact_lr=1.08e-5))


def ucf_128f(vtokens=True):
    """configs/ucf/mebt_128f.yaml — same network, block_size 8192, grid [32,16,16], dropout 0"""
    c = sky_16f(vtokens, dropout=0.0)
    p = c.model.params
    p.block_size, p.vis_epoch, p.t_prior = 8192, 50, "gaussian100000_2"
    m = c.model.mask.params
    m.max_token, m.shape, m.budget = 8192, [32, 16, 16], 8192
    c.data.sequence_length = 128
    c.exp = AttrDict(exact_lr=0.00003, weight_decay=0.0001)
    return c


def taichi_16f(vtokens=True):
    """configs/taichi/mebt_16f.yaml — the Sky network without dropout, t_prior longest, lr 3e-5 / wd 1e-4, batch 32"""
    c = sky_16f(vtokens, dropout=0.0)
    c.model.params.t_prior = "longest"
    c.data.batch_size, c.data.sample_every_n_frames = 32, 4
    c.exp = AttrDict(exact_lr=0.00003, weight_decay=0.0001)
    return c


def vqgan_args(n_hiddens=32, downsample=(4, 8, 8), embedding_dim=256, n_codes=16384, sequence_length=16, resolution=128):
    """first-stage hyper-parameters for BASELINE config 5 (TATS-style values, SURVEY.md §8f: they live in the first-stage
    checkpoint's hyper_parameters, not in this repository's YAMLs): [B,3,16,128,128] -> [B,4,16,16] tokens of 16384 codes"""
    import argparse
    return argparse.Namespace(n_hiddens=n_hiddens, downsample=tuple(downsample), image_channels=3, embedding_dim=embedding_dim,
                              n_codes=n_codes, norm_type="group", padding_type="replicate", sequence_length=sequence_length,
                              sample_every_n_frames=1, resolution=resolution)


def tiny(vtokens=True):
    """BASELINE.json configs[0]: n_layer=4, n_embd=256, block=256, sos_emb=64 on [B,2,8,8] tokens"""
    c = sky_16f(vtokens, dropout=0.0)
    p = c.model.params
    p.block_size, p.n_layer, p.n_head, p.n_embd, p.sos_emb = 256, 4, 4, 256, 64
    p.mode = ["latent_enc", "latent_self", "latent_dec", "lt2l"]
    m = c.model.mask.params
    m.max_token, m.shape, m.budget = 256, [2, 8, 8], 128
    return c


def forward_flops_per_sample(params, NC, NT, vocab=16384):
    """Algorithmic forward FLOPs of one sample (matmuls only, 2 FLOP/MAC): SURVEY.md §8d formula
    2 * [ sum_blocks ((2 NQ + 2 NK) d^2 + 2 NQ NK d + 8 NQ d^2) + NT d V ], (NQ, NK) per routing mode (gpt.py:164-179).
    `params` = the transformer config node (n_embd, sos_emb, mode)."""
    d, NS = params.n_embd, params.sos_emb
    total = 0
    for mode in params.mode:
        NQ, NK = {"latent_enc": (NS, NC), "latent_self": (NS, NS), "latent_dec": (NT, NS), "lt2l": (NS, NS + NT),
                  "maskgit": (NC + NT, NC + NT)}[mode]
        total += (2 * NQ + 2 * NK) * d * d + 2 * NQ * NK * d + 8 * NQ * d * d
    return 2 * (total + NT * d * vocab)


def build_model(cfg, compute_dtype="bf16", device=None):
    from .transformer import Net2NetTransformer
    model = Net2NetTransformer(cfg.model.params, cfg.model.vqvae, cfg.model.mask,
                               cond_stage_key=cfg.model.params.cond_stage_key)
    model.compute_dtype = compute_dtype
    model.learning_rate = cfg.exp.exact_lr                       # train_transformer.py:54
    model.warmup_steps = cfg.exp.get("warmup_steps", 0)           # :55-58
    model.weight_decay = cfg.exp.get("weight_decay", 0.01)        # :59-62
    model.cosine_lr = cfg.exp.get("cosine_lr", False)             # :63-66
    return model.to(device) if device is not None else model
