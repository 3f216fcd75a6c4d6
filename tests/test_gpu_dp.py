"""Data parallelism with the REAL HIP engine on one MI355X: two ranks share GPU 0 and talk over gloo (a 1-GPU box cannot
run RCCL between two ranks on the same device), each running `TrainLoop` with the bucketed reducer of mebt_amd/parallel.py:
the real `NativeModel` bucket ranges, `mebt_op_cast_bf16` -> reduce-scatter -> `mebt_adamw_slice` on the optimizer stream
-> all-gather of the bf16 mirror / fp32 tail (deferred: issued after backward, the next forward waits bucket by bucket), `consolidate()`.  2 ranks x batch 2 must equal 1 process x batch 4 after three
steps (the DDP contract, reference train_transformer.py:39-41).  GPU only."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import assert_same_trajectory

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _make(dtype):
    from mebt_amd import presets
    torch.manual_seed(5)
    cfg = presets.tiny()
    cfg.exp.exact_lr = 1e-3
    m = presets.build_model(cfg, compute_dtype=dtype)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.02)
    return m


def _batches(n=4):
    g = torch.Generator().manual_seed(12)
    xs = [torch.randint(0, 16384, (n, 2, 8, 8), generator=g) for _ in range(3)]
    idxs = [torch.stack([torch.randperm(128, generator=g) for _ in range(n)]) for _ in range(3)]
    return xs, idxs, (0.45, 0.3, 0.0)            # t = 0: NC = 0 (empty key/value reductions: zero-filled gradient slices)


def _worker(rank, world, port, dtype, mode, wire, defer, ret, overlap=True, check_buckets=True, total=4):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["MEBT_DP_DEFER_GATHER"] = "1" if defer else "0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mebt_amd.parallel import GradReducer
        from mebt_amd.trainer import TrainLoop
        model = _make(dtype).to(DEV).train()
        red = GradReducer(world_size=world, mode=mode, wire=wire, layers_per_bucket=2)
        loop = TrainLoop(model, red, overlap_optimizer=overlap)
        assert not loop.fused_optimizer and (loop.opt_stream is not None) == overlap
        # the engine's wire-gradient binding follows the path that will run: bound only for sharded bf16
        assert (loop.native.gWb is not None) == (mode == "sharded" and wire == "bf16" and dtype == "bf16")
        xs, idxs, ts = _batches(total)
        per = total // world
        # batches resident before the loop: a pageable host-to-device copy inside it waits for the whole device and would
        # serialise every step behind the optimizer stream's work
        xs = [x[rank * per:(rank + 1) * per].to(DEV) for x in xs]
        idxs = [i[rank * per:(rank + 1) * per].to(DEV) for i in idxs]
        for x, idx, t in zip(xs, idxs, ts):
            st = loop.step(x, idx, t=t)
            if mode == "sharded" and check_buckets:      # deferred: one event per bucket handed to the engine (head, blocks [3,2], [1], [0], non-Linear), none otherwise
                assert red.defer == defer and len(getattr(loop.native, "_fw_events", [])) == (5 if defer else 0)
                assert len(red._sharded_ranges) == (5 if world == 2 else 3)      # world 3: head and non-Linear bucket replicated
        # one tuner per job (VERDICT r03 #6a): only rank 0 times GEMM candidates in situ, the other ranks adopted its table after
        # steps 1 and 2 (TrainLoop) and never tuned themselves; after one more sync every rank holds rank 0's table
        from mebt_amd import _lib
        mine = _lib.tune_table_text()
        red.sync_tune_table()
        tabs = [None] * world
        dist.all_gather_object(tabs, (_lib.tune_table_text(), mine, bool(_lib.load().mebt_gemm_autotune_enabled())))
        assert all(t[0] == tabs[0][0] for t in tabs), "tune tables differ between ranks after the sync"
        assert tabs[0][2] and not any(t[2] for t in tabs[1:]), "only rank 0 may tune in situ"
        lead = set(tabs[0][0].splitlines())
        assert all(set(t[1].splitlines()) <= lead for t in tabs[1:]), "a follower held an entry rank 0 never chose"
        if dtype == "bf16":
            assert tabs[0][0].count("\n") > 3, tabs[0][0]     # version line + the signatures the tiny config tuned
        loss = red.mean_scalars(st[4:5].clone()).cpu()
        stale = red.master_stale
        if stale:
            with pytest.raises(RuntimeError, match="consolidate"):
                model.state_dict()
        loop.consolidate()
        torch.cuda.synchronize()
        if rank == 1:      # the non-zero rank: its fp32 masters of rank 0's shards came through consolidate()
            sd = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
            ret.put((sd, float(loss), stale, [a.cpu().numpy().copy() for a in loop.native.adam], red.bytes_on_wire))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,mode,wire,defer,world", [("f32", "sharded", "fp32", True, 2), ("bf16", "sharded", "bf16", True, 2),
                                                          ("bf16", "sharded", "bf16", False, 2), ("f32", "allreduce", "fp32", True, 2),
                                                          ("bf16", "sharded", "bf16", True, 3), ("bf16", "sharded", "bf16", True, -2)])
def test_two_ranks_on_one_gpu_equal_one_rank_double_batch(dtype, mode, wire, defer, world):
    # world = -2: two ranks WITHOUT an optimizer stream (MEBT_OVERLAP_OPT=0 / overlap_optimizer=False): the sharded calls run
    # in line on the compute stream, with the bf16 wire gradients bound (that combination used to update the Linear
    # weights from a never-written fp32 gradient buffer)
    overlap = world > 0
    world = abs(world)
    # world = 3: the head (16384 x 256) and the non-Linear bucket do not cut into 4 x 3 aligned shards and take the replicated
    # fallback — with the bf16 wire gradients bound, i.e. an all-reduce of the bf16 buffer the weight-gradient launches wrote
    import torch.multiprocessing as mp
    from mebt_amd.trainer import TrainLoop
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 32100 + (os.getpid() % 1500) + {"fp32": 0, "bf16": 3}[wire] + (5 if mode == "allreduce" else 0) + (7 if not defer else 0) + 13 * (world - 2) + (17 if not overlap else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, dtype, mode, wire, defer, ret, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    sd2, loss2, stale, adam2, wire_bytes = ret.get(timeout=900)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # one process, the whole batch, replicated optimizer
    model = _make(dtype).to(DEV).train()
    loop = TrainLoop(model, fused_optimizer=False)
    xs, idxs, ts = _batches()
    used = (4 // world) * world                 # the samples the ranks consumed
    for x, idx, t in zip(xs, idxs, ts):
        st = loop.step(x[:used].to(DEV), idx[:used].to(DEV), t=t)
    torch.cuda.synchronize()
    lr = 1e-3
    assert abs(float(st[4]) - loss2) < (1e-5 if dtype == "f32" else 2e-3) * abs(float(st[4]))
    assert stale == (mode == "sharded" and dtype == "bf16")
    n_w, n_p = loop.native.n_w, loop.native.n_p
    if mode == "sharded" and world == 2:
        gsz = 2 if wire == "bf16" else 4
        assert wire_bytes == 3 * ((n_w + n_p) * gsz + n_w * (2 if dtype == "bf16" else 4) + n_p * 4)     # three steps
    worst = 0.0
    for k, v in model.state_dict().items():
        d = np.abs(v.cpu().numpy() - sd2[k]).max()
        worst = max(worst, d)
        if dtype == "f32" and not k.endswith("attn.key.bias"):      # key.bias: zero gradient, AdamW amplifies rounding noise
            assert d <= 2e-5 * (1 + np.abs(sd2[k]).max()), (k, d)
        else:
            assert d <= 6.6 * lr, (k, d)                             # three +-lr steps of a ~0 gradient whose sign flipped
    for a, b in zip(loop.native.adam, adam2):
        ref = a.cpu().numpy()
        tol = (1e-4 if dtype == "f32" else 3e-2) * np.abs(ref).max()
        assert np.abs(ref - b).max() <= tol
    print(f"[dp {dtype} {mode} wire {wire} defer {defer}] max |dp| vs single process {worst:.3e}; loss {loss2:.5f}")


@pytest.mark.parametrize("dtype,wire", [("f32", "fp32"), ("bf16", "bf16")])
def test_two_ranks_on_one_gpu_equal_the_oracle_steps(dtype, wire):
    """The data-parallel HIP path against the CPU ORACLE, not against the single-process HIP step: two ranks
    sharing the GPU (batch 2 each, sharded reducer: all-to-all / reduce-scatter -> AdamW on the owned shard -> all-gather), three
    steps at t = 0.45, 0.3, 0.0, against three `oracle.train_step` on the four samples (reference train_transformer.py:39-41: DDP's
    mean gradient = the gradient of the mean loss).  fp32 engine: parameters to rounding; bf16 engine + bf16 wire: within the steps
    AdamW takes, loss to 2e-3."""
    import torch.multiprocessing as mp
    from mebt_amd.launch import free_port
    from oracle import mebt_oracle as orc
    from tests.test_gpu_benchsize import oracle_cfg_of
    from mebt_amd import presets
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dtype, "sharded", wire, True, ret, True)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        sd2, loss2, stale, adam2, _ = ret.get(timeout=300)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    cfg = presets.tiny()
    sd0 = {k: v.detach().clone() for k, v in _make(dtype).state_dict().items()}
    st = orc.TrainState(sd0, lr=1e-3)
    ocfg = oracle_cfg_of(cfg)
    xs, idxs, ts = _batches()
    for x, idx, t in zip(xs, idxs, ts):
        r = orc.train_step(st, ocfg, x, idx, t)
    lr = 1e-3
    assert abs(r["loss"] - loss2) < (2e-5 if dtype == "f32" else 2e-3) * abs(r["loss"]), (r["loss"], loss2)
    worst, off = 0.0, 0
    for k, ref in st.P.items():
        d = np.abs(sd2[k] - ref.detach().numpy())
        worst = max(worst, float(d.max()))
        assert d.max() <= 6.6 * lr, (k, float(d.max()))                 # three +-lr steps of a ~0 gradient whose sign flipped, at most
        if dtype == "f32" and not k.endswith("attn.key.bias"):
            tol = 5e-5 * (1.0 + float(np.abs(sd2[k]).max()))
            assert (d > tol).mean() < 2e-3, (k, float((d > tol).mean()), float(d.max()))
    print(f"[dp {dtype} sharded, wire {wire}] 2 ranks x batch 2 vs 3 oracle steps on 4 samples: max |dp| {worst:.3e}, loss {loss2:.5f} (oracle {r['loss']:.5f})")
    from tests.helpers import record_measured
    record_measured(f"dp 2 ranks sharing the GPU vs 3 oracle steps ({dtype}, wire {wire}): max |d parameter|", worst, 6.6 * lr)


# measured (4 ranks, tiny config, 3 steps): relative L2 distance between the AdamW first moments (= the averaged gradients' EMA) of a
# bf16-wire and an fp32-wire run: 5.1e-3 for the Linear weights (one bf16 rounding per rank's gradient + 3 on the wire; bf16 eps is
# 3.9e-3), 3.3e-3 for the non-Linear tail (fp32 on the wire in both runs: it only sees the weights drift); gate = 2 x measured
WIRE_BF16_VS_FP32_M_RELL2 = {4: 1e-2, 8: 1.6e-2}      # MEBT_DP_EXCHANGE=rs, 8 ranks: 7 roundings on the wire per element; gate = 2 x measured (see profiles/r04_parity_measured.txt)
# default exchange (all-to-all + fp32 sum of the ranks' bf16 copies inside the owner's AdamW launch): what is left is
# each rank's ONE rounding of its own contribution; gate = 2 x measured on MI355X (profiles/r06_parity_measured.txt)
A2A_VS_FP32_M_RELL2 = {4: 6e-3, 8: 4.5e-3}          # measured 2.93e-3 / 2.19e-3 (reduce-scatter: 5.11e-3 / 8.25e-3)


@pytest.mark.parametrize("ranks", [4, pytest.param(8, marks=pytest.mark.skipif(os.environ.get("MEBT_LONG_TESTS") == "0", reason="MEBT_LONG_TESTS=0: without the 8-rank case"))])
def test_bf16_wire_against_fp32_wire_at_four_ranks(ranks):
    """VERDICT r02 weak #11 / r05 missing #2: the reference's DDP sums fp32 gradients (train_transformer.py:39-41).  Four (eight)
    ranks sharing the GPU (batch 1 each), bf16 engine, sharded mode: the same three steps with the gradients exchanged (a) in bf16
    by all-to-all and summed in fp32 by the owner (the default), (b) reduced in bf16 by reduce-scatter (MEBT_DP_EXCHANGE=rs: N - 1
    roundings of partial sums per element), (c) reduced in fp32.  All runs keep the DDP contract against one process x batch N
    (same bounds as the two-rank test), and the reduced gradients themselves - seen through AdamW's first moments - differ from
    the fp32 run by measured, gated amounts; (a) must be closer to it than (b)."""
    import torch.multiprocessing as mp
    from mebt_amd.trainer import TrainLoop
    ctx = mp.get_context("spawn")
    runs = {}
    for arm, wire, exchange in (("a2a", "bf16", "a2a"), ("rs", "bf16", "rs"), ("fp32", "fp32", "rs")):
        from mebt_amd.launch import free_port
        ret = ctx.Queue()
        port = free_port()
        os.environ["MEBT_DP_EXCHANGE"] = exchange                 # inherited by the spawned ranks
        try:
            procs = [ctx.Process(target=_worker, args=(r, ranks, port, "bf16", "sharded", wire, True, ret, True, False, ranks)) for r in range(ranks)]
            for p in procs:
                p.start()
        finally:
            del os.environ["MEBT_DP_EXCHANGE"]
        try:
            runs[arm] = ret.get(timeout=600)
            for p in procs:
                p.join(timeout=120)
                assert p.exitcode == 0
        finally:
            for p in procs:                      # a rank that failed to start must not leave the others waiting for it
                if p.is_alive():
                    p.terminate()
    model = _make("bf16").to(DEV).train()
    loop = TrainLoop(model, fused_optimizer=False)
    xs, idxs, ts = _batches(ranks)
    for x, idx, t in zip(xs, idxs, ts):
        st = loop.step(x.to(DEV), idx.to(DEV), t=t)
    torch.cuda.synchronize()
    lr = 1e-3
    for arm, (sd, loss, stale, adam, _) in runs.items():
        assert abs(float(st[4]) - loss) < 2e-3 * abs(float(st[4])), (arm, loss)
        for k, v in model.state_dict().items():
            assert np.abs(v.cpu().numpy() - sd[k]).max() <= 6.6 * lr, (arm, k)
    from tests.helpers import record_measured
    worst = {}
    for arm in ("a2a", "rs"):
        rel = []
        for a, b in zip(runs[arm][3], runs["fp32"][3]):
            rel.append(float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30)))
        print(f"[dp wire bf16 ({arm}) vs fp32, {ranks} ranks] relative L2 of the optimizer state tensors {[f'{r:.2e}' for r in rel]}")
        worst[arm] = max(rel[0], rel[2] if len(rel) > 2 else 0.0)
    record_measured(f"dp_wire_bf16_vs_fp32_m_rell2[{ranks} ranks, reduce-scatter: sum in bf16]", worst["rs"], WIRE_BF16_VS_FP32_M_RELL2[ranks])
    record_measured(f"dp_wire_bf16_vs_fp32_m_rell2[{ranks} ranks, all-to-all: sum in fp32 (default)]", worst["a2a"], A2A_VS_FP32_M_RELL2[ranks])
    assert worst["rs"] <= WIRE_BF16_VS_FP32_M_RELL2[ranks], worst
    assert worst["a2a"] <= A2A_VS_FP32_M_RELL2[ranks] and worst["a2a"] < worst["rs"], worst


def _rccl_worker(port, dtype, mode, wire, delay, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if delay:      # ~2 ms (at 2.4 GHz) in front of every bucket's optimizer / gather work on the optimizer stream
        os.environ["MEBT_DP_TEST_DELAY_CYCLES"] = str(delay)
    torch.cuda.set_device(0)
    from mebt_amd.parallel import init_rccl
    init_rccl(0, 1, torch.device("cuda", 0))
    try:
        from mebt_amd.parallel import GradReducer
        from mebt_amd.trainer import TrainLoop
        model = _make(dtype).to(DEV).train()
        red = GradReducer(world_size=1, mode=mode, wire=wire, force=True)
        assert red.active and red._inplace
        loop = TrainLoop(model, red)
        assert not loop.fused_optimizer and loop.opt_stream is not None
        xs, idxs, ts = _batches()
        xs, idxs = [x.to(DEV) for x in xs], [i.to(DEV) for i in idxs]      # resident: no device-wide wait between steps
        for x, idx, t in zip(xs, idxs, ts):
            st = loop.step(x, idx, t=t)
        loss = red.mean_scalars(st[4:5].clone()).cpu()
        loop.consolidate()
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
        ret.put((sd, float(loss), red.bytes_on_wire, loop.native.gWb is not None))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,mode,wire,delay", [("bf16", "sharded", "bf16", 0), ("f32", "sharded", "fp32", 0), ("f32", "allreduce", "fp32", 0),
                                                    ("bf16", "sharded", "bf16", 5_000_000), ("f32", "sharded", "fp32", 5_000_000)])
def test_rccl_executes_the_collective_path_with_one_rank(dtype, mode, wire, delay):
    """RCCL itself (torch.distributed backend "nccl") on the one GPU of this box: a process group of size 1 with the
    reducer forced active, so that `reduce_scatter_tensor` (bf16 and fp32), the in-place `all_gather_into_tensor` into the
    bf16 mirror / fp32 tail, `all_reduce`, `broadcast`, the asynchronous work handles and the hand-offs between the compute,
    RCCL and optimizer streams all run exactly as in the N-GPU job (a shard is then the whole bucket).  Result == the plain
    single-process step.  `delay`: every piece of optimizer-stream work (shard AdamW, the deferred gathers) is held back by
    ~2 ms, so the host enqueues the next forward long before the parameters are final and the per-bucket waits
    (mebt_model_set_forward_waits) are what orders it: with the waits dropped (`set_forward_waits([])` after each step) the
    same run ends at loss 9.7744 instead of 9.7710 — stale weights — once the optimizer stream really runs beside the
    compute stream (TrainLoop picks it by probing, parallel.pick_concurrent_stream: streams that share a hardware queue
    execute in submission order and hide the race).  The batches are resident before the loop: a pageable host-to-device
    copy waits for the whole device."""
    import torch.multiprocessing as mp
    from mebt_amd.trainer import TrainLoop
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 33400 + (os.getpid() % 1500) + {"fp32": 0, "bf16": 3}[wire] + (5 if mode == "allreduce" else 0) + (9 if delay else 0)
    p = ctx.Process(target=_rccl_worker, args=(port, dtype, mode, wire, delay, ret))
    p.start()
    sd1, loss1, wire_bytes, wire_grads = ret.get(timeout=900)
    p.join(timeout=120)
    assert p.exitcode == 0
    model = _make(dtype).to(DEV).train()
    loop = TrainLoop(model, fused_optimizer=False)
    xs, idxs, ts = _batches()
    for x, idx, t in zip(xs, idxs, ts):
        st = loop.step(x.to(DEV), idx.to(DEV), t=t)
    torch.cuda.synchronize()
    assert wire_bytes > 0 and wire_grads == (dtype == "bf16" and mode == "sharded")
    assert abs(float(st[4]) - loss1) < (1e-6 if dtype == "f32" else 2e-3) * abs(float(st[4]))
    lr = 1e-3
    for k, v in model.state_dict().items():
        d = np.abs(v.cpu().numpy() - sd1[k]).max()
        if dtype == "f32" and not k.endswith("attn.key.bias"):
            assert_same_trajectory(v, sd1[k], k, lr=lr)                     # one rank: the "sum" is the gradient itself
        else:
            assert d <= 6.6 * lr, (k, d)                                      # bf16 wire: sign flips of ~0 gradients, three steps


def _rccl_soak_worker(port, ret):
    import random
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    from mebt_amd.parallel import GradReducer, init_rccl
    init_rccl(0, 1, torch.device("cuda", 0))
    try:
        from mebt_amd import presets
        from mebt_amd.trainer import TrainLoop
        cfg = presets.sky_16f(dropout=0.1)
        cfg.exp.exact_lr = 1e-4
        torch.manual_seed(3)
        model = presets.build_model(cfg, compute_dtype="bf16").to(DEV).train()
        red = GradReducer(world_size=1, force=True)            # defaults: sharded, bf16 wire, tapered buckets, deferred gathers
        loop = TrainLoop(model, red)
        assert red.active and red.defer and not loop.fused_optimizer and loop.native.gWb is not None
        g = torch.Generator().manual_seed(5)
        base = torch.randint(0, 32, (6, 1, 16, 16), generator=g)
        x = ((base + torch.arange(4).view(1, 4, 1, 1) * 7) % 16384).to(DEV)
        rng = random.Random(1)
        first = None
        for step in range(400):
            idx = torch.stack([torch.randperm(1024, generator=g) for _ in range(6)]).to(DEV)
            st = loop.step(x, idx, t=rng.random() * 0.98)
            if step == 0:
                first = float(st[4].cpu())
        last = float(st[4].cpu())
        model.eval()                                           # the inference forward honours the pending per-bucket waits too
        perm = torch.stack([torch.randperm(1024, generator=g) for _ in range(6)]).to(DEV)
        ci, ti = perm[:, :256].contiguous(), perm[:, 256:].contiguous()
        with torch.no_grad():
            logits, _ = model.reconstruct_mask(x, ci, ti)
        acc = float((logits.argmax(-1) == torch.gather(x.reshape(6, -1), 1, ti)).float().mean())
        loop.consolidate()
        finite = all(bool(torch.isfinite(v).all()) for v in model.state_dict().values())
        ret.put((first, last, acc, finite))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_path_trains_the_full_size_network_like_the_fused_path():
    """The data-parallel step as every rank runs it (bf16 gradients from the weight-gradient launches, reduce-scatter, AdamW on
    the shard, deferred all-gathers behind per-bucket forward waits; one rank, real RCCL) over 400 steps of the real training
    regime — dropout, a new permutation and t ~ U(0,1) per step — at the benchmarked size: the same memorisation the fused
    single-GPU path reaches (tests/test_gpu_fullsize.py), then inference on the gathered weights."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    p = ctx.Process(target=_rccl_soak_worker, args=(35200 + (os.getpid() % 1500), ret))
    p.start()
    first, last, acc, finite = ret.get(timeout=900)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert abs(first - np.log(16384)) < 0.6 and np.isfinite(last) and last < 0.5 and acc > 0.9 and finite, (first, last, acc, finite)
    print(f"[dp soak] loss {first:.3f} -> {last:.4f} in 400 sharded steps; reconstruction {100 * acc:.1f} %")


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO torch.distributed environment (the shape of the command the driver runs for N = 1):
    the parent starts the two ranks as a child job, relays the one JSON line and exits with the child's code.
    Both ranks share this box's GPU over gloo (MEBT_BENCH_SHARE_GPU=1): functional, not a performance number."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MEBT_BENCH_SHARE_GPU"] = "1"
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2", "--secondary", "none", "--no-cpu-baseline"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["value"] > 0
    dp = r["data_parallel"]
    assert dp["dp_mode"] == "sharded" and dp["wire"] == "bf16" and dp["rccl_ranks"] == 2 and dp["dp_fallback"] is None
    assert dp["bytes_on_wire_per_step"] > 1e9 and dp["scaling_efficiency"] > 0 and "exposed_comm_ms" in dp
    # the line judges itself against the N = 1 headline (the fused step) and carries what a first multi-GPU run needs for diagnosis
    assert abs(dp["scaling_efficiency"] - dp["one_rank_ms"]["fused_optimizer"] / r["ms_per_step"]) < 2e-3
    assert abs(dp["scaling_efficiency_vs_separate_optimizer"] - dp["one_rank_ms"]["separate_optimizer"] / r["ms_per_step"]) < 2e-3
    assert isinstance(dp["comm_env"], dict) and "rccl_version" in dp["comm_env"]
    waits = dp["exposed_forward_wait_ms_per_bucket"]        # gloo staging is synchronous: the keys exist, the waits are ~0
    assert waits is None or all(v >= 0 for v in waits.values())
    assert "cuda_initialized=False" in out.stderr


def test_bench_fallback_to_allreduce_runs_under_rccl():
    """bench.py's escape hatch (a sharded first step that raises -> `GradReducer.abandon()` -> model rebuilt -> legacy bucketed
    fp32 all-reduce with a replicated optimizer) executed ON RCCL before the scaling node ever needs it (VERDICT r03 #6c): a
    one-rank RCCL group with the reducer forced active (MEBT_DP_FORCE=1), the failure injected after a real, half-applied
    sharded step (MEBT_BENCH_FAIL_SHARDED=1).  The line must say so (`dp_fallback`), and the path it describes must be the
    all-reduce one."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update({"MEBT_DP_FORCE": "1", "MEBT_BENCH_FAIL_SHARDED": "1", "MASTER_PORT": str(36100 + os.getpid() % 1500)})
    out = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--secondary", "none", "--no-cpu-baseline"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    dp = r["data_parallel"]
    assert dp["dp_fallback"] and "injected failure" in dp["dp_fallback"]
    assert dp["dp_mode"] == "allreduce" and dp["wire"] == "fp32" and dp["backend"] == "nccl" and dp["rccl_ranks"] == 1
    assert r["value"] > 0 and r["config"]["optimizer"].startswith("replicated") and abs(r["config"]["loss"] - 9.7) < 0.5
    assert r["roofline"]["frac"] > 0
