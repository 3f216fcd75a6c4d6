#!/usr/bin/env python3
"""Headline benchmark of the MeBT hot path on MI355X (contract: see the task's bench.py section).

metric  : masked video tokens/sec/GPU, train step (BASELINE.json `metric`)
workload: Sky-Timelapse 16f config (BASELINE.json configs[1]): 24L / d=1024 / 16 heads, 1024 VQ
          tokens + 256 latents, batch 6 per GPU, t = 0.5 -> NC = NT = 512 (SURVEY.md §8d headline
          point), synthetic token grids, random-init weights.
step    : embed -> 24 blocks -> head -> masked-token CE (+top-1/5) -> full backward ->
          (all-reduce when N>1) -> fused AdamW.  Inputs are resident in HBM before timing starts.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` = the GEMM kernel family (93 % of the step's FLOPs),
achieved = algorithmic FLOPs / HIP-event time on the launch stream, measured live after the timed
region; `cpu_baseline` = the CPU oracle (kind "port") timed on this box's host cores on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def synthetic_batch(B, shape, rank, device):
    g = torch.Generator().manual_seed(1234 + rank)
    N = shape[0] * shape[1] * shape[2]
    x = torch.randint(0, 16384, (B, *shape), generator=g)
    idx = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
    return x.to(device), idx.to(device)


def cpu_baseline(model_sd, cfg, t, sample_B=6):
    """The CPU oracle (a port of the reference algorithm, pinned to it by tests/golden) timed on
    this box's host cores: one full train step (fwd + CE + bwd + AdamW) of the same network at
    batch `sample_B` and the same t."""
    from oracle import mebt_oracle as orc
    p = cfg.model.params
    ocfg = orc.OracleConfig(p.n_layer, p.n_head, p.n_embd, p.block_size, p.sos_emb, p.mode, shape=cfg.model.mask.params.shape,
                            schedule=cfg.model.mask.params.schedule, budget=cfg.model.mask.params.budget, avg_loss=1.0)
    cores = min(os.cpu_count(), 16)        # more threads than this only oversubscribes these op sizes
    torch.set_num_threads(cores)
    st = orc.TrainState({k: v.float().cpu() for k, v in model_sd.items()}, lr=cfg.exp.exact_lr)
    x, idx = synthetic_batch(sample_B, cfg.model.mask.params.shape, 0, "cpu")
    orc.train_step(st, ocfg, x, idx, t)                      # untimed warm-up step (allocator, thread pool)
    dts = []
    for _ in range(3):                                       # ~12-15 s of CPU work in all
        t0 = time.perf_counter()
        r = orc.train_step(st, ocfg, x, idx, t)
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[1]                                      # median of three
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": r["n_targets"] / dt, "unit": "masked tokens/s", "cores": cores, "kind": "port",
            "sample": f"median of 3 timed train steps (fwd+CE+bwd+AdamW, after 1 warm-up) at batch {sample_B}, NC=NT={r['n_targets'] // sample_B}, fp32, "
                      f"torch {torch.__version__} CPU, {torch.get_num_threads()} threads, {model}; {dt:.1f} s per step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=6)
    ap.add_argument("--t", type=float, default=0.5)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preset", default="sky_16f", choices=["sky_16f", "tiny"])
    args = ap.parse_args()

    import torch.distributed as dist
    from mebt_amd import presets
    from mebt_amd.parallel import GradReducer
    from mebt_amd.trainer import TrainLoop
    from mebt_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = "nccl"                                      # RCCL
    if os.environ.get("MEBT_BENCH_SHARE_GPU") == "1":     # functional check of the N>1 path on a 1-GPU box: all ranks on GPU 0, gloo
        local_rank, backend = 0, "gloo"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=device if backend == "nccl" else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    # the Sky config trains with embd/resid/attn dropout 0.1 (configs/stl/mebt_16f.yaml:12-14): the measured
    # step includes it (counter-based masks, recomputed in backward)
    cfg = presets.sky_16f(vtokens=True, dropout=args.dropout) if args.preset == "sky_16f" else presets.tiny()
    torch.manual_seed(0)                       # identical random-init weights on every rank
    model = presets.build_model(cfg, compute_dtype=args.dtype).to(device).train()
    loop = TrainLoop(model, GradReducer(world_size=world))
    shape = cfg.model.mask.params.shape
    x, idx = synthetic_batch(args.batch, shape, rank, device)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        stats = loop.step(x, idx, t=args.t)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = loop.step(x, idx, t=args.t)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    stats = stats.cpu()
    n_targets = int(stats[3])                                  # masked tokens scored per rank per step
    ms = 1e3 * elapsed / args.steps
    value = world * n_targets * args.steps / elapsed

    # dominant kernel family: MFMA GEMMs — time every launch of 2 more steps with HIP events on the
    # launch stream (rank 0 records; every rank runs the steps because they contain collectives)
    lib = _lib.load()
    if rank == 0:
        lib.mebt_profile_enable(1)
    for _ in range(2):
        loop.step(x, idx, t=args.t)
    sync()
    roof = None
    import ctypes as C
    n, tms, fl = C.c_double(), C.c_double(), C.c_double()
    if rank == 0:
        _lib.check(lib.mebt_profile_read(0, C.byref(n), C.byref(tms), C.byref(fl)))
        nb, tb, by = C.c_double(), C.c_double(), C.c_double()
        _lib.check(lib.mebt_profile_read(1, C.byref(nb), C.byref(tb), C.byref(by)))
        lib.mebt_profile_enable(0)
    # the same two steps with every launch on one stream (identical to the above unless MEBT_SIDE_STREAM=1)
    ser = None
    lib.mebt_debug_side_stream(loop.native.h, 0)
    if rank == 0:
        lib.mebt_profile_enable(1)
    for _ in range(2):
        loop.step(x, idx, t=args.t)
    sync()
    if rank == 0:
        n2, t2, f2 = C.c_double(), C.c_double(), C.c_double()
        _lib.check(lib.mebt_profile_read(0, C.byref(n2), C.byref(t2), C.byref(f2)))
        lib.mebt_profile_enable(0)
        ser = round(f2.value / (t2.value * 1e-3) / 1e12, 2) if t2.value > 0 else None
    lib.mebt_debug_side_stream(loop.native.h, 1 if loop.native.side_stream else 0)
    if rank == 0:
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        # HBM-side bytes per GEMM launch from the committed rocprofv3 PMC passes of this same command
        # (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction): tools/pmc_traffic.py
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                traffic = round(json.load(f)["gemm_bf16"]["hbm_bytes_per_launch"]) if args.dtype == "bf16" else None
        except (OSError, KeyError, ValueError):
            pass
        achieved = fl.value / (tms.value * 1e-3) / 1e12 if tms.value > 0 else 0.0
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": round(by.value / max(1.0, nb.value)),
                "kernel": "bf16 MFMA GEMM family (gemm_bf16_dma[_ks2] / gemm_pair / wgrad_grouped incl. its fused AdamW epilogue)" if args.dtype == "bf16" else "gemm_f32_kernel",
                "launches_per_step": n.value / 2, "gemm_ms_per_step": round(tms.value / 2, 3),
                "gemm_gflop_per_step": round(fl.value / 2 / 1e9, 1), "achieved_single_stream": ser}

    if rank == 0:
        out = {"metric": "masked video tokens/sec/GPU (train step, 24L d=1024, 1024+256 tok)",
               "value": round(value, 1), "unit": "masked tokens/s (whole job)", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "per_gpu": round(value / world, 1),
               "config": {"workload": "Sky-Timelapse 16f MeBT train step: 24L/1024d/16h, 1024 VQ tokens + 256 latents, "
                                      f"batch {args.batch}/GPU, t={args.t} (NC=NT={n_targets // args.batch}), "
                                      "fwd + masked CE + bwd + AdamW" + (" + RCCL all-reduce" if world > 1 else ""),
                          "global_batch": args.batch * world, "parallelism": f"dp{world}", "dropout": args.dropout,
                          "loss": round(float(stats[4]), 4)},
               "roofline": roof}
        if not args.no_cpu_baseline and world == 1:
            sd = {k: v.detach() for k, v in model.state_dict().items()}
            out["cpu_baseline"] = cpu_baseline(sd, cfg, args.t)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
