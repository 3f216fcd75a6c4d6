"""CPU tier: host logic of the product (config loading, MaskGen bookkeeping, module/param schema,
optimizer groups, LR schedule), the C-ABI library's symbol table, and the data-parallel reducer over
gloo with world_size 2.  No compute call into the HIP library happens here (no GPU)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from oracle import mebt_oracle as orc
from tests.golden import make_golden as mg
from tests.helpers import load_golden, product_config

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from mebt_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mebt_hip.h")).read()
    declared = set(re.findall(r"\b(mebt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(_lib.PROTOTYPES) == declared, set(_lib.PROTOTYPES) ^ declared      # the binding covers the whole header
    bound = _lib.load()
    assert bound.mebt_abi_version() == 2


def test_shipped_tune_table_matches_this_build():
    """mebt_amd/tune/gfx950.txt (tools/make_tune_table.sh) is merged at load time so that a fresh process — every data-parallel
    rank, every bench run — does not stall on in-situ GEMM tuning.  A table written by a build with other
    variant codes is ignored as a whole (MEBT_TUNE_VERSION): this test fails when the version was bumped without regenerating it."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from mebt_amd import _lib; t = _lib.tune_table_text(); s = open(_lib.SHIPPED_TUNE).read(); "
            "print(len(t.splitlines()), len(s.splitlines()), t.splitlines()[0] == s.splitlines()[0])")
    env = {k: v for k, v in os.environ.items() if k not in ("MEBT_GEMM_TUNE_CACHE", "MEBT_GEMM_TUNE_SHIPPED")}
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    loaded, shipped, same_version = out.stdout.split()
    assert same_version == "True" and int(loaded) == int(shipped) > 500, out.stdout
    env["MEBT_GEMM_TUNE_SHIPPED"] = "0"                     # fresh-tuning runs start from an empty table
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.split()[0] == "1", out.stdout


def test_tune_cache_of_another_version_is_moved_aside_not_truncated(tmp_path):
    """MEBT_GEMM_TUNE_CACHE pointing at a table of another MEBT_TUNE_VERSION (or at a file that is no table at all): the file is kept
    as <cache>.v<its version>, a fresh versioned table starts by rename, under a lock several ranks take (the old code
    truncated the user's file in place, without a lock)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cache = tmp_path / "tune.txt"
    old = "4 3 1536 1024 1024 5\n4 3 3072 1024 1024 7\n"                       # unversioned: reads as version 1
    cache.write_text(old)
    code = "from mebt_amd import _lib; print(_lib.tune_table_text().splitlines()[0])"
    env = dict(os.environ, MEBT_GEMM_TUNE_CACHE=str(cache), MEBT_GEMM_TUNE_SHIPPED="0")
    procs = [subprocess.Popen([sys.executable, "-c", code], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(3)]
    outs = [p_.communicate(timeout=300) for p_ in procs]
    assert all(p_.returncode == 0 for p_ in procs), [o[1][-500:] for o in outs]
    version = outs[0][0].strip()
    assert version.startswith("1 -1 ") and all(o[0].strip() == version for o in outs)
    assert (tmp_path / "tune.txt.v1").read_text() == old                      # nothing of the user's file was lost
    assert cache.read_text().splitlines() == [version]                        # one version line, however many processes raced
    assert sum("kept as" in o[1] for o in outs) == 1


def test_product_fails_loudly_without_gpu():
    """no CPU fallback: a forward on a CPU-resident model must raise, not silently compute"""
    from tests.helpers import build_product
    if torch.cuda.is_available():
        pytest.skip("GPU visible")
    model = build_product("micro", "f32", device="cpu")
    x, idx = mg.inputs("micro", 2, "cpu")
    with pytest.raises(RuntimeError):
        model(x, None, t=0.5, indices=idx)


def test_config_loader_and_instantiate(tmp_path):
    from mebt_amd.config import load_config, AttrDict, instantiate_from_config
    y = tmp_path / "c.yaml"
    y.write_text("model:\n  target: mebt.transformer.Net2NetTransformer\n  params:\n    n_layer: 24\n    mode: [latent_enc, latent_self]\n"
                 "  mask:\n    target: tats.mask_sampler.MaskGen\n    params: {schedule: linear, shape: [4, 16, 16], budget: 1024}\nexp:\n  exact_lr: 1.08e-5\n")
    cfg = load_config([str(y)], ["model.params.n_layer=4", "exp.warmup_steps=10", "--data.batch_size=6"])
    assert cfg.model.params.n_layer == 4 and cfg["exp"]["warmup_steps"] == 10 and cfg.data.batch_size == 6
    assert "cosine_lr" not in cfg.exp and not hasattr(cfg.exp, "cosine_lr") and cfg.exp.get("x", 3) == 3
    assert abs(cfg.exp.exact_lr - 1.08e-5) < 1e-12
    ms = instantiate_from_config(cfg.model.mask)           # legacy 'tats.' prefix is rewritten (reference utils.py:6)
    assert type(ms).__name__ == "MaskGen" and ms.schedule == "linear" and ms.budget == 1024
    import utils as top_level_utils                       # the import path the reference's scripts use
    assert top_level_utils.instantiate_from_config is instantiate_from_config


def test_maskgen_divide_indices_matches_reference_golden():
    from mebt.mask_sampler import MaskGen
    g = load_golden("divide_indices")
    idx = torch.from_numpy(g["indices"])
    for k, (sched, num) in enumerate(zip(g["case_sched"], g["case_num"])):
        t, T, start, training, budget, seq_len = num
        ms = MaskGen(schedule=str(sched), shape=(4, 2, 2), budget=int(budget))
        ms.train(bool(training))
        oc, orr = np.random.choice, np.random.randint
        np.random.choice = lambda a, p=None, _T=int(T): _T
        np.random.randint = lambda lo, hi=None, _s=int(start): _s
        try:
            c, tg, sl = ms.divide_indices(idx, torch.tensor(float(t)), np.arange(4) + 1, np.ones(4))
        finally:
            np.random.choice, np.random.randint = oc, orr
        assert sl == int(seq_len)
        assert c.shape == g[f"k{k}_ctx"].shape and (c.numpy() == g[f"k{k}_ctx"]).all(), (k, sched, num)
        assert tg.shape == g[f"k{k}_tgt"].shape and (tg.numpy() == g[f"k{k}_tgt"]).all(), (k, sched, num)
    with pytest.raises(ValueError):
        MaskGen(schedule="nope")


def test_module_schema_optimizer_groups_and_lr_schedule():
    from mebt.transformer import Net2NetTransformer
    tcfg, vcfg, mcfg = product_config("micro")
    model = Net2NetTransformer(tcfg, vcfg, mcfg, cond_stage_key="label")
    shapes = orc.param_shapes(mg.oracle_cfg("micro"))
    sd = model.state_dict()
    assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
    g = load_golden("train_micro")
    model.learning_rate, model.weight_decay = 1e-3, 0.05
    opt = model.configure_optimizers()
    assert [len(grp["params"]) for grp in opt.param_groups] == list(g["group_sizes"])
    assert [grp["weight_decay"] for grp in opt.param_groups] == list(g["group_wd"])
    decay, emb, no_decay, pos = orc.decay_split({k: None for k in shapes})
    named = {id(p): n for n, p in model.named_parameters()}
    assert sorted(named[id(p)] for p in opt.param_groups[0]["params"]) == sorted("transformer." + n for n in decay)
    # warm-up / cosine (reference transformer.py:665-678)
    model.warmup_steps, model.cosine_lr, model.trainer.max_steps = 10, True, 110
    for step in (0, 5, 9, 10, 60, 110):
        model.trainer.global_step = step
        assert abs(model.learning_rate * model.lr_scale() - orc.lr_at(step, 1e-3, 10, True, 110)) < 1e-12
    # constructible from the shipped-style config with vtokens False (token grids only)
    tcfg2, vcfg2, mcfg2 = product_config("micro", vtokens=False)
    m2 = Net2NetTransformer(tcfg2, vcfg2, mcfg2)
    with pytest.raises(NotImplementedError):
        m2.encode_to_z(torch.zeros(1, 3, 4, 8, 8))
    # unsupported routing fails loudly
    tcfg3, vcfg3, mcfg3 = product_config("micro_maskgit")
    m3 = Net2NetTransformer(tcfg3, vcfg3, mcfg3)
    assert [b.mode for b in m3.transformer.blocks] == ["latent_enc", "latent_dec", "maskgit"]   # padding rule gpt.py:208-209


def _dp_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mebt_amd.parallel import GradReducer
        from mebt_amd.engine import flat_layout
        torch.set_num_threads(1)
        cfg = mg.oracle_cfg("micro")
        P = orc.closed_form_params(cfg)
        shapes = orc.param_shapes(cfg)
        x, idx = mg.inputs("micro", 4, "dp")
        per = 4 // world
        xs, ids = x[rank * per:(rank + 1) * per], idx[rank * per:(rank + 1) * per]

        class FakeNative:            # the reducer only touches the flat gradient buffers and the bucket ranges
            n_layer, n_embd = cfg.n_layer, cfg.n_embd
            def layer_w_range(self, hi, lo):
                per_l = 12 * self.n_embd * self.n_embd
                return lo * per_l, (hi + 1) * per_l
            def head_w_range(self):
                return self.n_layer * 12 * self.n_embd * self.n_embd, self.gW.numel()
            def layer_p_range(self, hi, lo):
                return lo * 13 * self.n_embd, (hi + 1) * 13 * self.n_embd
            def tail_p_range(self):
                return self.n_layer * 13 * self.n_embd, self.gP.numel()
        nat = FakeNative()
        Wn, Pn = flat_layout(cfg.n_layer)
        st = orc.TrainState(P, lr=1e-3, weight_decay=0.05)
        red = GradReducer(world_size=world)

        def hook(grads):             # flatten exactly like the engine lays the buffers out, reduce bucket by bucket
            nat.gW = torch.cat([grads["transformer." + n if not n.startswith("transformer.") else n].reshape(-1) for n in Wn])
            nat.gP = torch.cat([grads[n].reshape(-1) for n in Pn])
            red.bucket_ready(nat, "head", None, None)
            hi = cfg.n_layer - 1
            while hi >= 0:
                lo = max(0, hi - 1)
                red.bucket_ready(nat, "layers", hi, lo)
                hi = lo - 1
            red.bucket_ready(nat, "embed", None, None)
            red.wait()
            for flat, names in ((nat.gW, Wn), (nat.gP, Pn)):
                off = 0
                for n in names:
                    k = int(np.prod(shapes[n]))
                    grads[n] = (flat[off:off + k] * red.grad_scale).view(shapes[n]).clone()
                    off += k
        r = orc.train_step(st, cfg, xs, ids, 0.4, grad_hook=hook)
        t = red.mean_scalars(torch.tensor([r["loss"]], dtype=torch.float64))
        # one GEMM tuner per job (VERDICT r03 #6a): followers switch their in-situ tuner off and adopt rank 0's table at the sync points
        from mebt_amd import _lib
        lib = _lib.load()
        assert lib.mebt_gemm_autotune_enabled() == 1
        red.lead_tuning()
        assert lib.mebt_gemm_autotune_enabled() == (1 if rank == 0 else 0)
        version = _lib.tune_table_text().splitlines()[0]           # "1 -1 <MEBT_TUNE_VERSION>": texts of another version are ignored
        assert _lib.tune_table_merge(f"{version}\n4 3 {1536 + 128 * rank} 1024 1024 {100673538 + rank}\n", replace=True) == 1
        assert _lib.tune_table_merge("4 3 1536 1024 1024 5\n", overwrite=True) == 0      # unversioned text: another build's codes
        text = red.sync_tune_table()
        assert text == _lib.tune_table_text() and "4 3 1536 1024 1024 100673538" in text and "1664" not in text
        # the text only travels when rank 0's table changed since the last sync: 8 bytes otherwise
        assert red.sync_tune_table() is None
        if rank == 0:
            assert _lib.tune_table_merge(f"{version}\n4 3 2048 1024 1024 100673539\n", overwrite=True) == 1
        text2 = red.sync_tune_table()
        assert text2 == _lib.tune_table_text() and "4 3 2048 1024 1024 100673539" in text2
        # a FOLLOWER whose table was changed locally is re-synced too, and only then
        if rank == 1:
            assert _lib.tune_table_merge(f"{version}\n4 3 4096 1024 1024 100673540\n", overwrite=True) == 1
        text3 = red.sync_tune_table()
        assert text3 == text2 == _lib.tune_table_text() and "4096 1024 1024 100673540" not in _lib.tune_table_text()
        assert red.sync_tune_table() is None
        assert [red.tune_sync_due(k) for k in (16, 32, 64, 96, 128, 192)] == [True, True, True, False, True, False]
        assert [red.tune_sync_due(k) for k in (0, 1, 2, 3, 4, 5, 8, 255, 256, 257, 512, 640, 768)] == [False, True, True, False, True, False, True, False, True, False, True, False, True]
        if rank == 0:
            ret.put(({k: v.detach().numpy().copy() for k, v in st.P.items()}, float(t)))   # numpy: pickled by value
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_data_parallel_reducer_gloo_world2():
    """2 ranks x half batch with bucketed all-reduce == 1 process x full batch (the DDP contract,
    reference train_transformer.py:39-41): parameters after one AdamW step and the mean loss."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    P2, loss2 = ret.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = mg.oracle_cfg("micro")
    x, idx = mg.inputs("micro", 4, "dp")
    st = orc.TrainState(orc.closed_form_params(cfg), lr=1e-3, weight_decay=0.05)
    r = orc.train_step(st, cfg, x, idx, 0.4)
    assert abs(r["loss"] - loss2) < 1e-5 * abs(r["loss"])        # mean of per-rank mean losses == full-batch loss (equal shards)
    for k, v in st.P.items():
        assert np.allclose(v.detach().numpy(), P2[k], rtol=2e-4, atol=2e-6), k


def test_token_dataset_matches_reference_items(tmp_path):
    """mebt_amd.data.TokenClipDataset == reference HDF5Dataset_vtokens (mebt/data.py:330-414) item by item for the same
    torch seed: clip start, resampling of too-short videos, spatial crop box, frame skipping, `indices` permutation."""
    import torch
    from mebt_amd.data import TokenClipDataset
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_contract.npz"))
    f = str(tmp_path / "tokens.npz")
    np.savez(f, train_data=g["train_data"], train_idx=g["train_idx"], test_data=g["test_data"], test_idx=g["test_idx"])
    cases = [("full", dict(sequence_length=4, resolution=6, spatial_length=6, sample_every_n_frames=1, latent_shape=[4, 6, 6])),
             ("crop", dict(sequence_length=4, resolution=6, spatial_length=4, sample_every_n_frames=1, latent_shape=[4, 4, 4])),
             ("skip", dict(sequence_length=6, resolution=6, spatial_length=6, sample_every_n_frames=2, latent_shape=[3, 6, 6]))]
    for tag, kw in cases:
        for train in (True, False):
            ds = TokenClipDataset(f, train=train, **kw)
            pre = f"{tag}_{'train' if train else 'test'}"
            assert len(ds) == g[pre + "_video"].shape[0]
            torch.manual_seed(123)
            for i in range(len(ds)):
                it = ds[i]
                assert it["video"].dtype == torch.int64
                assert np.array_equal(it["video"].numpy(), g[pre + "_video"][i]), (pre, i)
                assert np.array_equal(it["indices"].numpy(), g[pre + "_indices"][i]), (pre, i)
                box = np.asarray(it["cbox"]).reshape(-1) if not np.isscalar(it["cbox"]) else np.zeros(4, np.int64)
                assert np.array_equal(box, g[pre + "_cbox"][i]), (pre, i)


def test_sharded_sampler_equals_distributed_sampler():
    """ShardedSampler == torch DistributedSampler (what the reference's VideoData uses, data.py:271-275) for every rank,
    several world sizes / epochs, with and without shuffle / drop_last."""
    import torch
    from torch.utils.data.distributed import DistributedSampler
    from mebt_amd.data import ShardedSampler, TokenData, SyntheticTokenDataset
    for n in (1, 7, 16, 37):
        ds = list(range(n))
        for world in (1, 2, 3, 8):
            for shuffle in (True, False):
                for drop_last in (False, True):
                    for epoch in (0, 3):
                        for rank in range(world):
                            ref = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle, seed=5, drop_last=drop_last)
                            ref.set_epoch(epoch)
                            mine = ShardedSampler(n, world, rank, shuffle=shuffle, seed=5, drop_last=drop_last)
                            mine.set_epoch(epoch)
                            assert list(ref) == list(mine) and len(ref) == len(mine), (n, world, shuffle, drop_last, epoch, rank)
    # the data module yields the {'video','indices'} batches the training step consumes
    from mebt_amd.config import AttrDict
    dl = TokenData(AttrDict(latent_shape=[2, 4, 4], batch_size=3, num_workers=0), world_size=2, rank=1).train_dataloader()
    b = next(iter(dl))
    assert b["video"].shape == (3, 2, 4, 4) and b["indices"].shape == (3, 32) and b["video"].dtype == torch.int64
    assert sorted(b["indices"][0].tolist()) == list(range(32))


def _dp_sharded_worker(rank, world, port, ret, wire, exchange="a2a"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mebt_amd.parallel import GradReducer
        from mebt_amd.engine import flat_layout
        torch.set_num_threads(1)
        cfg = mg.oracle_cfg("micro")
        P0 = orc.closed_form_params(cfg)
        shapes = orc.param_shapes(cfg)
        x, idx = mg.inputs("micro", 6, "dp6")
        per = 6 // world
        xs, ids = x[rank * per:(rank + 1) * per], idx[rank * per:(rank + 1) * per]
        Wn, Pn = flat_layout(cfg.n_layer)
        lr, wd = 1e-3, 0.05

        class FakeNative:            # the flat-buffer interface the reducer drives, with torch ops in place of the HIP kernels
            n_layer, n_embd = cfg.n_layer, cfg.n_embd
            def __init__(self):
                self.W = torch.cat([P0[n].reshape(-1) for n in Wn]).clone()
                self.P = torch.cat([P0[n].reshape(-1) for n in Pn]).clone()
                self.Wlp = self.W.to(torch.bfloat16)
                self.gW, self.gP = torch.zeros_like(self.W), torch.zeros_like(self.P)
                self.adam = [torch.zeros_like(self.W), torch.zeros_like(self.W), torch.zeros_like(self.P), torch.zeros_like(self.P)]
            def layer_w_range(self, hi, lo):
                per_l = 12 * self.n_embd * self.n_embd
                return lo * per_l, (hi + 1) * per_l
            def head_w_range(self):
                return self.n_layer * 12 * self.n_embd * self.n_embd, self.gW.numel()
            def layer_p_range(self, hi, lo):
                return lo * 13 * self.n_embd, (hi + 1) * 13 * self.n_embd
            def tail_p_range(self):
                return self.n_layer * 13 * self.n_embd, self.gP.numel()
            def cast_bf16(self, src, dst):
                dst.copy_(src)
            def sync_lowp(self, force=False):
                self.Wlp.copy_(self.W)
            def adamw_slice(self, which, off, n, grad, lr, weight_decay, step, betas=(0.9, 0.95), eps=1e-8, grad_scale=1.0, stream=None, pieces=1):
                p = (self.W if which == 0 else self.P)[off:off + n]
                m_, v_ = self.adam[2 * which][off:off + n], self.adam[2 * which + 1][off:off + n]
                if pieces > 1:       # mebt_adamw_slice_pieces: the ranks' bf16 copies, added in fp32 in rank order
                    assert grad.dtype == torch.bfloat16 and grad.numel() == pieces * n
                    self.pieces_seen = pieces
                    grad = grad.view(pieces, n).float().sum(0)
                orc.adamw_update(p, grad.float() * grad_scale, m_, v_, step, lr, weight_decay if which == 0 else 0.0, betas[0], betas[1], eps)
                if which == 0:
                    self.Wlp[off:off + n].copy_(p)
        nat = FakeNative()
        red = GradReducer(world_size=world, mode="sharded", wire=wire, layers_per_bucket=2, exchange=exchange)
        red.broadcast_parameters(nat)
        Pg = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
        logits, z_t, ntw, seq_len = orc.forward(Pg, cfg, xs, ids, 0.4, training=True)
        _, _, loss = orc.loss_and_acc(logits, z_t, ntw, seq_len, cfg)
        loss.backward()
        nat.gW = torch.cat([Pg[n].grad.reshape(-1) for n in Wn])
        nat.gP = torch.cat([Pg[n].grad.reshape(-1) for n in Pn])
        red.reduce_update(nat, "head", None, None, lr, wd, 1)
        plan = red.bucket_plan(cfg.n_layer)
        assert plan == [2, 2, 1, 1] and sum(plan) == cfg.n_layer       # tapered: the exposed last bucket is the smallest
        hi = cfg.n_layer - 1
        for size in plan:
            lo = hi - size + 1
            red.reduce_update(nat, "layers", hi, lo, lr, wd, 1)
            hi = lo - 1
        red.reduce_update(nat, "embed", None, None, lr, wd, 1)
        red.finish()
        stages = [("head", None, None)] + [("layers", h, h - s + 1) for h, s in zip(np.cumsum([cfg.n_layer - 1] + [-s for s in plan[:-1]]).tolist(), plan)] + [("embed", None, None)]
        all_ranges = [r for st in stages for r in red.sharded_ranges(nat, *st)]
        assert sum(b - a for w, a, b in all_ranges) == nat.W.numel() + nat.P.numel()              # the buckets tile both buffers exactly
        divisible = sum(1 for w, a, b in all_ranges if (b - a) % (4 * world) == 0)
        # between steps: the bf16 mirror and P are complete everywhere, the fp32 master of W only on its owners
        mirror_ok = bool(torch.isfinite(nat.Wlp.float()).all())
        stale = red.master_stale
        W_before = nat.W.clone()
        red.consolidate(nat, optimizer_state=True)
        changed = int((nat.W != W_before).sum())
        assert (nat.Wlp.float() - nat.W).abs().max() <= 8e-3 * nat.W.abs().max()      # mirror == bf16(master) after the gather
        assert getattr(nat, "pieces_seen", 1) == (world if (wire == "bf16" and exchange == "a2a") else 1)
        if rank == 1:
            ret.put((nat.W.numpy().copy(), nat.P.numpy().copy(), nat.adam[0].numpy().copy(), nat.adam[3].numpy().copy(), stale,
                     changed, mirror_ok, red.bytes_on_wire, len(red._sharded_ranges), divisible, len(all_ranges)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("wire,world,exchange", [("fp32", 2, "rs"), ("bf16", 2, "a2a"), ("bf16", 2, "rs"), ("fp32", 3, "rs"), ("bf16", 3, "a2a")])
def test_data_parallel_sharded_optimizer_gloo(wire, world, exchange):
    """reduce-scatter -> AdamW on the owned shard -> all-gather (the default data-parallel path, parallel.py) on 2 ranks x
    half batch == 1 process x full batch with the replicated optimizer: parameters and moments after one step, read on the
    NON-zero rank after consolidate().  fp32 wire: to rounding; bf16 wire: every parameter within the +-lr step AdamW takes
    at step 1, moments to bf16 resolution.  world = 3: the head (16384 x 64), the per-layer non-Linear slices and the
    tail do not divide into aligned shards and take the replicated fallback (all-reduce + full-range update) next to the
    sharded W buckets.  bf16 wire, exchange "a2a" (default): all-to-all of the shards and the fp32 sum of the ranks' bf16 copies
    inside the owner's AdamW (mebt_adamw_slice_pieces) - one bf16 rounding per rank, none of the sum's; "rs": RCCL-style
    reduce-scatter adding in bf16."""
    import torch.multiprocessing as mp
    from mebt_amd.engine import flat_layout
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + (7 if wire == "bf16" else 0) + 11 * world + (3 if exchange == "a2a" else 0)
    procs = [ctx.Process(target=_dp_sharded_worker, args=(r, world, port, ret, wire, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    W2, P2, mW2, vP2, stale, changed, mirror_ok, wire_bytes, n_sharded, divisible, n_ranges = ret.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = mg.oracle_cfg("micro")
    x, idx = mg.inputs("micro", 6, "dp6")
    st = orc.TrainState(orc.closed_form_params(cfg), lr=1e-3, weight_decay=0.05)
    r = orc.train_step(st, cfg, x, idx, 0.4)
    Wn, Pn = flat_layout(cfg.n_layer)
    Wref = np.concatenate([st.P[n].detach().numpy().reshape(-1) for n in Wn])
    Pref = np.concatenate([st.P[n].detach().numpy().reshape(-1) for n in Pn])
    mWref = np.concatenate([st.m[n].numpy().reshape(-1) for n in Wn])
    vPref = np.concatenate([st.v[n].numpy().reshape(-1) for n in Pn])
    assert stale and changed > 0 and mirror_ok            # rank 1 did NOT hold rank 0's shards of the fp32 master before consolidate
    n_all = W2.size + P2.size
    if world == 2:
        assert wire_bytes == (2 if wire == "bf16" else 4) * n_all + 2 * W2.size + 4 * P2.size    # reduce-scatter + all-gather payloads
        assert n_sharded == n_ranges == 6     # head + 4 W layer buckets (2, 2, 1, 1 layers) + one bucket of all non-Linear parameters
    else:
        assert n_sharded == divisible == 4 and n_ranges == 6      # only the four W layer buckets divide by 4 x 3; the head and the non-Linear bucket went the replicated way
    if wire == "fp32":
        assert np.allclose(W2, Wref, rtol=2e-4, atol=2e-6) and np.allclose(P2, Pref, rtol=2e-4, atol=2e-6)
        assert np.allclose(mW2, mWref, rtol=1e-4, atol=1e-6 * np.abs(mWref).max()) and np.allclose(vP2, vPref, rtol=1e-4, atol=1e-6 * np.abs(vPref).max())
    else:
        assert np.abs(W2 - Wref).max() <= 2.2e-3 and np.abs(P2 - Pref).max() <= 2.2e-3
        assert (np.abs(W2 - Wref) > 2e-5).mean() < 0.02                     # sign flips of ~0 gradients only
        assert np.abs(mW2 - mWref).max() <= 1.2e-2 * np.abs(mWref).max()
        # relative L2 error of the first moments (= 0.1 x the summed gradient at step 1) against the fp32 run: with the fp32 sum of
        # the all-to-all only each rank's own rounding of its contribution is left (<= 2^-9 relative per element, independent
        # across ranks); the bf16 reduce-scatter rounds the partial sums again
        rel = float(np.linalg.norm(mW2 - mWref) / np.linalg.norm(mWref))
        print(f"[dp {wire} {exchange} world {world}] rel-L2 of the first moments vs the fp32 step: {rel:.3e}")
        assert rel <= (2.0e-3 if exchange == "a2a" else 4.0e-3), rel


def test_load_from_lightning_format_checkpoint(tmp_path):
    """A checkpoint laid out as pytorch_lightning writes it for the reference module (download.py:56-61): `state_dict` with the
    reference's parameter names, `hyper_parameters` = the constructor arguments recorded by save_hyperparameters()
    (transformer.py:146) as PLAIN nested dicts, plus Lightning's bookkeeping keys.  load_from_checkpoint rebuilds the module
    (mode list, vocabulary, mask sampler) and loads every tensor."""
    import torch
    from mebt.transformer import Net2NetTransformer
    from oracle import closed_form as cf
    cfg = mg.oracle_cfg("micro")
    sd = {k: torch.from_numpy(v) for k, v in cf.state_dict_numpy(orc.param_shapes(cfg)).items()}
    c = mg.CONFIGS["micro"]
    hp = {"transformer_config": {"unconditional": True, "vocab_size": 16384, "first_stage_vocab_size": 16384, "block_size": c["block_size"],
                                 "n_layer": c["n_layer"], "n_head": c["n_head"], "n_embd": c["n_embd"], "n_unmasked": 0, "embd_pdrop": 0.0,
                                 "resid_pdrop": 0.1, "attn_pdrop": 0.0, "sample_every_n_latent_frames": 0, "first_stage_key": "video",
                                 "cond_stage_key": "label", "vtokens": True, "vtokens_pos": False, "vis_epoch": 100, "sos_emb": c["sos_emb"],
                                 "avg_loss": True, "mode": list(c["mode"]), "class_cond_dim": None},
          "first_stage_config": {"params": {"ckpt_path": None, "ignore_keys": ["loss"]}},
          "mask_config": {"target": "mebt.mask_sampler.MaskGen",
                          "params": {"iid": False, "schedule": "linear", "max_token": c["block_size"], "method": "mlm", "shape": c["shape"],
                                     "t_range": [0.0, 1.0], "budget": c["budget"]}},
          "ckpt_path": None, "ignore_keys": [], "first_stage_key": "video", "cond_stage_key": "label", "pkeep": 1.0, "sos_token": 0}
    path = str(tmp_path / "epoch=3-step=1200.ckpt")
    torch.save({"epoch": 3, "global_step": 1200, "pytorch-lightning_version": "1.5.4", "state_dict": sd, "hyper_parameters": hp,
                "optimizer_states": [], "lr_schedulers": [], "callbacks": {}, "loops": {}}, path)
    m = Net2NetTransformer.load_from_checkpoint(path)
    assert [b.mode for b in m.transformer.blocks] == list(c["mode"]) and m.config.resid_pdrop == 0.1
    assert m.global_step == 1200 and m.current_epoch == 3 and m.mask_sampler.budget == c["budget"]
    got = m.state_dict()
    assert set(got) == set(sd)
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    # a checkpoint our own launcher writes round-trips through the same entry point
    path2 = str(tmp_path / "ours.ckpt")
    torch.save({"state_dict": m.state_dict(), "hyper_parameters": m.hparams, "global_step": 7}, path2)
    m2 = Net2NetTransformer.load_from_checkpoint(path2)
    assert m2.global_step == 7 and all(torch.equal(m2.state_dict()[k], v) for k, v in sd.items())


@pytest.mark.parametrize("flavour", ["plain", "attributedict"])
def test_load_reference_checkpoint_with_omegaconf_hparams(tmp_path, flavour):
    """a checkpoint as the REFERENCE's run writes it — `hyper_parameters` pickles OmegaConf DictConfig / ListConfig
    nodes (train_transformer.py:27-33, transformer.py:146), here produced by tests/golden/make_lightning_ckpt.py with stand-in
    classes under the real module / class names and OmegaConf's pickle layout.  It loads with NEITHER omegaconf NOR
    pytorch_lightning importable, through `mebt.load_transformer(gpt_ckpt, vqgan_ckpt=None)` exactly as
    draft_and_revise_videos.py:138 calls it, and a file naming any other foreign class is refused."""
    import pickle
    import torch
    from oracle import closed_form as cf
    from tests.golden import make_lightning_ckpt as mk
    cfg = mg.oracle_cfg("micro")
    sd = {k: torch.from_numpy(v) for k, v in cf.state_dict_numpy(orc.param_shapes(cfg)).items()}
    c = mg.CONFIGS["micro"]
    hp = {"transformer_config": {"unconditional": True, "vocab_size": 16384, "first_stage_vocab_size": 16384, "block_size": c["block_size"],
                                 "n_layer": c["n_layer"], "n_head": c["n_head"], "n_embd": c["n_embd"], "n_unmasked": 0, "embd_pdrop": 0.0,
                                 "resid_pdrop": 0.1, "attn_pdrop": 0.0, "sample_every_n_latent_frames": 0, "first_stage_key": "video",
                                 "cond_stage_key": "label", "vtokens": True, "vtokens_pos": False, "vis_epoch": 100, "sos_emb": c["sos_emb"],
                                 "avg_loss": True, "mode": list(c["mode"]), "class_cond_dim": None, "t_prior": "gaussian2"},
          "first_stage_config": {"params": {"ckpt_path": None, "ignore_keys": ["loss"]}},
          "mask_config": {"target": "mebt.mask_sampler.MaskGen",
                          "params": {"iid": False, "schedule": "linear", "max_token": c["block_size"], "method": "mlm", "shape": list(c["shape"]),
                                     "t_range": [0.0, 1.0], "budget": c["budget"]}},
          "ckpt_path": None, "ignore_keys": [], "first_stage_key": "video", "cond_stage_key": "label", "pkeep": 1.0, "sos_token": 0}
    path = mk.write(str(tmp_path / "epoch=11-step=50000.ckpt"), sd, hp, flavour)
    assert "omegaconf" not in sys.modules and "pytorch_lightning" not in sys.modules
    with pytest.raises(ModuleNotFoundError):                     # the file really does name those classes
        torch.load(path, map_location="cpu", weights_only=False)
    import mebt
    from mebt.download import load_transformer
    assert load_transformer is mebt.load_transformer
    m = load_transformer(path, vqgan_ckpt=None)
    assert not m.training and m.global_step == 50000 and m.current_epoch == 11
    assert [b.mode for b in m.transformer.blocks] == list(c["mode"]) and m.config.resid_pdrop == 0.1 and m.config.t_prior == "gaussian2"
    assert list(m.mask_sampler.shape) == list(c["shape"]) and m.mask_sampler.budget == c["budget"] and m.hparams["pkeep"] == 1.0
    assert type(m.config.mode) is list and type(m.config.n_layer) is int and m.config.class_cond_dim is None
    got = m.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], v) for k, v in sd.items())
    # what our launcher re-saves is plain containers: loads back the ordinary way too
    torch.save({"state_dict": m.state_dict(), "hyper_parameters": m.hparams, "global_step": 3}, str(tmp_path / "resaved.ckpt"))
    assert mebt.load_transformer(str(tmp_path / "resaved.ckpt"), None, torch.device("cpu")).global_step == 3

    class Evil:                                                   # any other foreign global is refused, not imported
        def __reduce__(self):
            return (os.system, ("true",))
    bad = str(tmp_path / "bad.ckpt")
    torch.save({"state_dict": {}, "hyper_parameters": {"x": Evil()}}, bad)
    with pytest.raises(pickle.UnpicklingError):
        mebt.load_transformer(bad)
    # pickle protocol 4 resolves DOTTED names inside an allow-listed module — ('torch', 'os.system') must not
    # reach os.system through torch's own `import os`; neither may a module of this package lend its imports
    from mebt_amd.lightning_shim import TolerantUnpickler
    import io

    def stack_global(module, name, arg):        # PROTO 4 | module | name | STACK_GLOBAL | arg | TUPLE1 | REDUCE | STOP
        def u(sv):
            b = sv.encode()
            return b"\x8c" + bytes([len(b)]) + b
        return b"\x80\x04" + u(module) + u(name) + b"\x93" + u(arg) + b"\x85R."
    marker = tmp_path / "pwned"
    for module, name in (("torch", "os.system"), ("torch", "serialization.os.system"), ("mebt_amd.launch", "subprocess.getoutput"),
                         ("numpy", "os.system"), ("functools", "partial"), ("builtins", "eval"), ("builtins", "getattr"),
                         ("torch.storage", "_load_from_bytes"), ("os", "system"), ("pathlib", "os.system")):
        with pytest.raises(pickle.UnpicklingError):
            TolerantUnpickler(io.BytesIO(stack_global(module, name, f"touch {marker}"))).load()
    assert not marker.exists()


def test_train_launcher_gpus_argument_follows_lightning():
    """`--gpus` as Lightning reads it (reference scripts/train_config_log_gpus.sh passes `--gpus 0,1,2,3,`): N, -1, lists, a single
    id with a trailing comma, and composition with an already exported HIP_VISIBLE_DEVICES."""
    from mebt_amd.train import parse_gpus
    assert parse_gpus(None, visible="") == (1, None)
    assert parse_gpus("8", visible="") == (8, None)
    assert parse_gpus("1", visible="") == (1, None) and parse_gpus("0", visible="") == (1, None)
    assert parse_gpus("-1", visible="", device_count=8) == (8, None)
    assert parse_gpus("-1", visible="4,5") == (2, None)
    assert parse_gpus("0,1,2,3,", visible="") == (4, ["0", "1", "2", "3"])
    assert parse_gpus("3,", visible="") == (1, ["3"])                      # Lightning: device 3, not "3 devices"
    assert parse_gpus("1,0", visible="4,6,7") == (2, ["6", "4"])            # indices into the visible set
    with pytest.raises(ValueError):
        parse_gpus("0,3", visible="4,6,7")
    with pytest.raises(ValueError):
        parse_gpus("4", visible="4,6,7")


def test_bench_parent_starts_ranks_as_a_child_and_never_touches_the_gpu(tmp_path):
    """`bench.py --gpus N` / `python -m mebt_amd.train --gpus 0,1` without a torch.distributed environment re-launch themselves
    through mebt_amd/launch.py: child process (never exec), stdout relayed, the child's return code, and the parent's
    torch.cuda never initialised.  The child command is replaced by a stub here (no GPU in the build container)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from mebt_amd.launch import launch_command
    cmd = launch_command(8, "bench.py", ["--gpus", "8"], port=1234)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd and cmd[-3:] == ["bench.py", "--gpus", "8"]
    assert "127.0.0.1" in cmd
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    stub = tmp_path / "child.py"
    stub.write_text("import sys\nprint('{\"value\": 1}')\nsys.exit(int(sys.argv[1]))\n")
    for script in (["bench.py", "--gpus", "2", "--steps", "1"], ["-m", "mebt_amd.train", "--preset", "tiny", "--gpus", "0,1"]):
        for code in (0, 3):
            env["MEBT_LAUNCH_CHILD"] = f"{sys.executable} {stub} {code}"
            out = subprocess.run([sys.executable] + script, cwd=root, env=env, capture_output=True, text=True, timeout=300)
            assert out.returncode == code, (script, out.stderr[-2000:])
            assert out.stdout.strip() == '{"value": 1}', out.stdout
            assert "cuda_initialized=False" in out.stderr and "starting 2 ranks" in out.stderr
    # inside a distributed launch (WORLD_SIZE / RANK set) nothing is spawned
    from mebt_amd import launch
    os.environ["WORLD_SIZE"], os.environ["RANK"] = "2", "0"
    try:
        assert launch.spawn_ranks_if_needed(2, "bench.py", []) is None
    finally:
        del os.environ["WORLD_SIZE"], os.environ["RANK"]
    assert launch.spawn_ranks_if_needed(1, "bench.py", []) is None


def test_epoch_hooks_and_video_logger_host_side(tmp_path):
    """Lightning's loop hooks the reference overrides (transformer.py:332-351) exist on the mirror; off the `vis_epoch` cadence, or
    without a first stage / logger, `on_validation_epoch_start` returns before anything touches the GPU (the reference dereferences
    both unconditionally); `VideoLogger` stores `[N, T, C, H, W]` videos as uint8 files and scalars as text."""
    import warnings
    from tests.helpers import product_config
    from mebt.transformer import Net2NetTransformer
    from mebt_amd.lightning_shim import VideoLogger
    tcfg, fscfg, mcfg = product_config("micro", vis_epoch=5)
    model = Net2NetTransformer(tcfg, fscfg, mcfg, cond_stage_key="label")
    assert model.vis_epoch == 5 and model.logger is None
    assert model.on_train_epoch_start() is None
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model.current_epoch = 0
        model.on_validation_epoch_start()                    # (0 + 1) % 5 != 0: nothing to do, no warning
    model.current_epoch = 4
    with pytest.warns(UserWarning, match="visualisation skipped"):
        model.on_validation_epoch_start()                    # on the cadence, but no first stage / logger
    log = VideoLogger(str(tmp_path / "v"))
    vid = torch.linspace(0, 1, 2 * 3 * 3 * 4 * 4).reshape(2, 3, 3, 4, 4)
    log.experiment.add_video("sample/x", vid, 7, fps=20)
    log.experiment.add_scalar("val/loss", 1.25, 7)
    log.experiment.flush()
    (tag, path, step, fps), = log.videos
    arr = np.load(path)
    assert (tag, step, fps) == ("sample/x", 7, 20) and arr.dtype == np.uint8 and arr.shape == (2, 3, 3, 4, 4)
    assert np.abs(arr.astype(np.float64) / 255 - vid.numpy()).max() <= 0.5 / 255 + 1e-7
    assert (tmp_path / "v" / "scalars.tsv").read_text().split() == ["val/loss", "7", "1.25"]


def test_every_environment_knob_is_documented():
    """INTEGRATION.md's table of environment knobs lists every MEBT_* variable the library, the Python layer and bench.py read
    (documentation drift)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    used = set()
    files = glob.glob(os.path.join(root, "mebt_amd", "**", "*"), recursive=True) + [os.path.join(root, "bench.py")]
    for f in files:
        if f.endswith((".py", ".hip", ".cpp", ".h")):
            with open(f, errors="ignore") as fh:
                used |= set(re.findall(r'(?:getenv\(|environ\.get\(|environ\[|environ\.setdefault\()\s*"(MEBT_[A-Z0-9_]+)"', fh.read()))
    assert len(used) > 30
    with open(os.path.join(root, "INTEGRATION.md")) as fh:
        doc = fh.read()
    missing = sorted(v for v in used if v not in doc)
    assert not missing, missing


def test_documents_point_at_files_and_tests_that_exist():
    """DESIGN / README / BASELINE / INTEGRATION cite evidence by path and by test name: every `profiles/...` file, `tools/...` script,
    `tests/...py` file and `test_...` function they name exists (prefixes such as `profiles/r02_` are allowed when something matches)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tests = set()
    for f in glob.glob(os.path.join(root, "tests", "*.py")):
        with open(f) as fh:
            tests |= set(re.findall(r"def (test_[A-Za-z0-9_]+)", fh.read()))
    bad = []
    for doc in ("DESIGN.md", "README.md", "BASELINE.md", "INTEGRATION.md"):
        with open(os.path.join(root, doc)) as fh:
            text = fh.read()
        for m in set(re.findall(r"profiles/[A-Za-z0-9_.\-]+", text)):
            m = m.rstrip(".")
            if not glob.glob(os.path.join(root, m + "*")):
                bad.append((doc, m))
        for m in set(re.findall(r"tools/[A-Za-z0-9_.\-]+\.(?:py|sh|hip)", text)) | set(re.findall(r"tests/[A-Za-z0-9_/.\-]+\.py", text)):
            if not os.path.exists(os.path.join(root, m)):
                bad.append((doc, m))
        for m in set(re.findall(r"\b(test_[a-z0-9_]+)\b", text)):
            if m.endswith("_") or m.startswith(("test_gpu_", "test_host_", "test_oracle_")):
                continue
            if m not in tests and not any(t.startswith(m) for t in tests):
                bad.append((doc, m))
    assert not bad, bad


def test_t_priors_and_beta_schedule_match_reference_golden():
    """Host logic of the training-time draws: the video-length priors (reference transformer.py:24-49) at several global
    steps, and the (alpha, beta) the beta(t) schedule hands to torch's Beta at given global steps (:229-241) — values
    recorded from the imported reference (tests/golden/edit_beta_priors.npz)."""
    import torch.distributions.beta as tdb
    from mebt_amd import transformer as T
    from tests.helpers import product_config
    from mebt.transformer import Net2NetTransformer
    g = np.load(os.path.join(G, "edit_beta_priors.npz"))
    for name in ("uniform", "gaussian2", "gaussian100000_2", "longest"):
        got = np.stack([T.T_PRIORS[name](g["prior_lengths"], int(s)) for s in g["prior_steps"]])
        np.testing.assert_allclose(got, g["prior_" + name], rtol=1e-12, atol=0)
    tcfg, vcfg, mcfg = product_config("micro", beta_params=[3.0, 9.0], beta_iter=1000)
    model = Net2NetTransformer(tcfg, vcfg, mcfg, cond_stage_key="label").train()
    assert model.beta and model.beta_iter == 1000.0
    calls, forced = [], list(g["beta_forced_t"])

    class FakeBeta:
        def __init__(self, a, b):
            calls.append((float(a), float(b)))

        def sample(self):
            return torch.tensor(forced[len(calls) - 1])
    real = tdb.Beta
    tdb.Beta = torch.distributions.beta.Beta = FakeBeta
    try:
        ts = []
        for gs in g["beta_gsteps"]:
            model.global_step = int(gs)
            ts.append(model._draw_t(False))
    finally:
        tdb.Beta = torch.distributions.beta.Beta = real
    np.testing.assert_allclose(np.array(calls), g["beta_calls"], rtol=0, atol=0)
    np.testing.assert_allclose(np.array(ts), g["beta_forced_t"], rtol=1e-7)
    # eval mode never uses the schedule (:240-241): a python-RNG draw
    model.eval()
    assert 0.0 <= model._draw_t(False) < 1.0 and len(calls) == len(g["beta_gsteps"])


def test_mebt_utils_alias_matches_reference_golden():
    """`mebt.utils.shift_dim` / `accuracy` (reference mebt/utils.py:30-53,80-94) under the reference's import
    path, against values from the imported reference."""
    import mebt.utils as mu
    g = np.load(os.path.join(G, "utils.npz"))
    x = torch.from_numpy(g["x"])
    for i, (a, b) in enumerate(g["shift_cases"]):
        y = mu.shift_dim(x, int(a), int(b))
        assert tuple(y.shape) == tuple(g[f"shift{i}"].shape) and torch.equal(y.contiguous(), torch.from_numpy(g[f"shift{i}"]))
    a1, a5 = mu.accuracy(torch.from_numpy(g["acc_logits"]), torch.from_numpy(g["acc_target"]), topk=(1, 5))
    np.testing.assert_allclose([float(a1), float(a5)], g["acc"], rtol=1e-6)
    import utils as top                                     # the top-level `utils.instantiate_from_config` of the YAML `target:` strings
    assert callable(top.instantiate_from_config)


@pytest.mark.skipif(not os.path.isdir("/root/reference/mebt"), reason="needs the reference checkout (build container only)")
def test_golden_fixtures_regenerate_bit_identically(tmp_path):
    """`python tests/golden/make_golden.py` as committed — ONE process, every generator — reproduces every
    committed fixture array for array (the recipe used to crash between two generators)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MEBT_GOLDEN_OUT=str(tmp_path))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "make_golden.py")], cwd=root, env=env,
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    assert made == sorted(f for f in os.listdir(G) if f.endswith('.npz')), made
    for f in made:
        a, b = np.load(os.path.join(tmp_path, f)), np.load(os.path.join(G, f))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (f, k)


def test_host_layer_under_asan_ubsan():
    """SURVEY §5 "sanitizers" / VERDICT r02 #14: the sequencing + C-ABI layer (engine.cpp, capi.cpp, errors.cpp — workspace carving,
    offset tables, argument validation) built with host-side AddressSanitizer + UBSan (`make -C mebt_amd/csrc asan`) and driven
    through its host-only paths in a subprocess (tests/asan_host_driver.py; LD_PRELOAD = clang's asan runtime, MEBT_HOST_ONLY=1).
    A finding aborts the subprocess.  (GPU sanitizers are not available on this pool; the device code is the regular build.)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mk = subprocess.run(["make", "-C", os.path.join(root, "mebt_amd", "csrc"), "-j8", "asan"], capture_output=True, text=True, timeout=1500)
    assert mk.returncode == 0, mk.stdout[-2000:] + mk.stderr[-2000:]
    rt = subprocess.run(["/opt/rocm/bin/hipcc", "--print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.isfile(rt), rt
    env = dict(os.environ, MEBT_HOST_ONLY="1", LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "asan_host_driver.py"), os.path.join(root, "mebt_amd", "lib", "libmebt_hip_asan.so"), root],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "no sanitizer finding" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_sampling_command_lines_parse_the_shipped_scripts_flags(tmp_path):
    """`python -m mebt_amd.sample` / `python -m mebt_amd.draft_and_revise` take the flags of the reference's two scripts
    (sample_vqgan_transformer_videos.py:160-193, draft_and_revise_videos.py:64-96) and name their outputs like them: the command
    lines of scripts/valid_dnr_config_ckpt_exp_ucf_128f.sh:10-15,30-35 parse unchanged, and the file names are the ones that script's
    next stage (`--np_draft ...`, `measure_fvd_with_numpy.py --np_file ...`) expects."""
    from mebt_amd import sample, draft_and_revise
    from mebt_amd.scripts_common import resolve_checkpoint
    line = ("--base cfg.yaml --gpt_ckpt g.ckpt --exp_name EXP --vid_c_temp 2.0 --total_length 128 --vid_n_steps 32 --context_size 128 --step_size 128 "
            "--verbose --dataset ucf101 --no_phase --n_sample 512 --run 3 --batch_size 4 --save_videos --decoding_strategy maskgit --top_k 32 "
            "--save_codemap --bootstrap 64 --save_n 20")
    a, unknown = sample.build_parser().parse_known_args(line.split())
    assert not unknown and a.bootstrap == 64 and a.top_k == 32 and a.temp == 1.0 and a.schedule == "cosine" and a.batch_size == 4
    assert resolve_checkpoint(a) == "g.ckpt" and a.save == "results/EXP"
    save_dir, save_np = sample.output_names(a)
    assert save_np == "results/EXP/numpy_files_128/ucf101/VID_n_steps32_k32_temp1.0_ctemp2.0linear_maskgit_cosine_no_phase_run3"
    assert save_dir == "results/EXP/videos_128/ucf101/VID_n_steps32_k32_temp1.0_ctemp2.0linear_maskgit_cosine_no_phase_run3"
    a.no_phase = False
    with pytest.raises(AssertionError):                       # the reference asserts --no_phase (:235)
        sample.output_names(a)
    draft_file = tmp_path / "VID_n_steps32_k32_temp1.0_ctemp2.0linear_maskgit_cosine_no_phase_run3_codemap.npy"
    np.save(str(draft_file), np.zeros((4, 32, 16, 16), dtype=np.int64))
    line = (f"--base cfg.yaml --gpt_ckpt g.ckpt --exp_name EXP --total_length 128 --n_revise 32 --M 2 --revise_t 0.1 --np_draft {draft_file} "
            "--context_size 128 --step_size 128 --verbose --dataset ucf101 --no_phase --n_sample 512 --run 3 --batch_size 4 --save_videos --save_n 20")
    b, unknown = draft_and_revise.build_parser().parse_known_args(line.split())
    assert unknown == ["--no_phase"]                          # the shipped script passes it; the reference's parser leaves it to OmegaConf's dotlist too
    resolve_checkpoint(b)
    draft, postfix = draft_and_revise.apply_np_draft(b)
    assert draft.shape == (4, 32, 16, 16) and b.n_draft == 32 and b.draft_t == 0.0 and postfix == "_ctemp2.0"
    _, save_np = draft_and_revise.output_names(b, postfix)
    assert save_np == "results/EXP/numpy_files_128/ucf101/VID_dnr_nd32_dt0.0_nr32_rt0.1_M2_ctemp2.0_run3"      # the --np_file of the script's FVD stage (:37)
    b.exp_name, b.gpt_ckpt = "nope", ""
    with pytest.raises(FileNotFoundError):
        resolve_checkpoint(b)
