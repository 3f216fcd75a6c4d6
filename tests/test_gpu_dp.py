"""Data parallelism with the REAL HIP engine on one MI355X: two ranks share GPU 0 and talk over gloo (a 1-GPU box cannot
run RCCL between two ranks on the same device), each running `TrainLoop` with the bucketed reducer of mebt_amd/parallel.py:
the real `NativeModel` bucket ranges, `mebt_op_cast_bf16` -> reduce-scatter -> `mebt_adamw_slice` on the optimizer stream
-> all-gather of the bf16 mirror / fp32 tail, `consolidate()`.  2 ranks x batch 2 must equal 1 process x batch 4 after three
steps (the DDP contract, reference train_transformer.py:39-41).  GPU only."""
import os

import numpy as np
import pytest
import torch

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


def _batches():
    g = torch.Generator().manual_seed(12)
    xs = [torch.randint(0, 16384, (4, 2, 8, 8), generator=g) for _ in range(3)]
    idxs = [torch.stack([torch.randperm(128, generator=g) for _ in range(4)]) for _ in range(3)]
    return xs, idxs, (0.45, 0.3, 0.0)            # t = 0: NC = 0 (empty key/value reductions: zero-filled gradient slices)


def _worker(rank, world, port, dtype, mode, wire, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mebt_amd.parallel import GradReducer
        from mebt_amd.trainer import TrainLoop
        model = _make(dtype).to(DEV).train()
        red = GradReducer(world_size=world, mode=mode, wire=wire, layers_per_bucket=2)
        loop = TrainLoop(model, red)
        assert not loop.fused_optimizer
        xs, idxs, ts = _batches()
        per = 4 // world
        for x, idx, t in zip(xs, idxs, ts):
            st = loop.step(x[rank * per:(rank + 1) * per].to(DEV), idx[rank * per:(rank + 1) * per].to(DEV), t=t)
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


@pytest.mark.parametrize("dtype,mode,wire", [("f32", "sharded", "fp32"), ("bf16", "sharded", "bf16"), ("f32", "allreduce", "fp32")])
def test_two_ranks_on_one_gpu_equal_one_rank_double_batch(dtype, mode, wire):
    import torch.multiprocessing as mp
    from mebt_amd.trainer import TrainLoop
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 32100 + (os.getpid() % 1500) + {"fp32": 0, "bf16": 3}[wire] + (5 if mode == "allreduce" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dtype, mode, wire, ret)) for r in range(2)]
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
    for x, idx, t in zip(xs, idxs, ts):
        st = loop.step(x.to(DEV), idx.to(DEV), t=t)
    torch.cuda.synchronize()
    lr = 1e-3
    assert abs(float(st[4]) - loss2) < (1e-5 if dtype == "f32" else 2e-3) * abs(float(st[4]))
    assert stale == (mode == "sharded" and dtype == "bf16")
    n_w, n_p = loop.native.n_w, loop.native.n_p
    if mode == "sharded":
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
    print(f"[dp {dtype} {mode} wire {wire}] max |dp| vs single process {worst:.3e}; loss {loss2:.5f}")
