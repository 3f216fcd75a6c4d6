#!/usr/bin/env python3
"""Training launcher — counterpart of reference train_transformer.py for the token-grid path.

  python -m mebt_amd.train --base cfg.yaml [more.yaml] [key.sub=value ...] --max_steps 1000
  python -m mebt_amd.train --base cfg.yaml --gpus 0,1,2,3,4,5,6,7      (the reference's flag, train_transformer.py:39,46: this
                                                                        process starts one rank per listed device as a child job)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m mebt_amd.train --base cfg.yaml

Same config handling as the reference (YAML list + dot-list overrides, train_transformer.py:25-27;
exp.{exact_lr,warmup_steps,weight_decay,cosine_lr}, :45-66), seed 42 on every rank (:11) so all ranks
draw the same `t`; one process per GPU with RCCL all-reduce (DDP, :39-41).  Data: token grids from a
`.npz` / `.h5` token file with the reference's `{train,test}_data` / `_idx` keys (`--tokens` or `data.data_path`,
mebt_amd/data.py) or synthetic grids; every item carries `indices = randperm(T*H*W)` like the reference datasets
(mebt/data.py:85,233,413,471); ranks read disjoint shards (DistributedSampler semantics).
Checkpoints use the Lightning layout {'state_dict','hyper_parameters','global_step','epoch'} plus 'mebt_amd_loop'
(AdamW moments, step counters, RNG states); `--ckpt_path` RESUMES from them (weights only when the file has no loop state).
"""
import argparse
import os
import random
import time

import numpy as np
import torch


def parse_gpus(spec, visible=None, device_count=None):
    """Lightning's `--gpus` (reference scripts/train_config_log_gpus.sh: `--gpus 0,1,2,3,`): an int N = the first N devices,
    -1 = all of them, a comma list (a trailing comma makes a single id a list: "3," is device 3) = exactly those.  Returns
    (number of ranks, device mask or None).  Listed ids index the devices this process can see: with HIP_VISIBLE_DEVICES
    already exported they select FROM that mask (as Lightning's indices do), and an id outside it raises instead of being
    silently ignored."""
    if spec is None or str(spec).strip() == "":
        return 1, None
    spec = str(spec).strip()
    visible = os.environ.get("HIP_VISIBLE_DEVICES") if visible is None else visible
    vis = [v for v in visible.split(",") if v != ""] if visible else None
    if "," not in spec:
        n = int(spec)
        if n == -1:
            if vis is not None:
                return max(1, len(vis)), None
            if device_count is None:
                import torch
                device_count = torch.cuda.device_count()       # does not initialise the GPU
            return max(1, int(device_count)), None
        if n < 0:
            raise ValueError(f"--gpus {spec}: expected N >= 0, -1, or a comma-separated device list")
        if vis is not None and n > len(vis):
            raise ValueError(f"--gpus {n} but HIP_VISIBLE_DEVICES={visible} exposes {len(vis)} devices")
        return max(1, n), None
    ids = [g.strip() for g in spec.split(",") if g.strip() != ""]
    if not ids:
        return 1, None
    if vis is not None:
        try:
            ids = [vis[int(i)] for i in ids]
        except IndexError:
            raise ValueError(f"--gpus {spec} selects a device outside HIP_VISIBLE_DEVICES={visible}") from None
    return len(ids), ids


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", nargs="*", default=[], metavar="base_config.yaml")
    ap.add_argument("--preset", default=None, help="sky_16f | ucf_128f | tiny (instead of --base)")
    ap.add_argument("--tokens", default=None, help=".npz / .h5 token file (train_data [frames,H,W], train_idx), see mebt_amd/data.py")
    ap.add_argument("--max_steps", type=int, default=100)
    ap.add_argument("--log_every", type=int, default=10)
    ap.add_argument("--ckpt_every", type=int, default=0)
    ap.add_argument("--default_root_dir", default="runs")
    ap.add_argument("--ckpt_path", default=None)
    ap.add_argument("--accumulate_grad_batches", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--gpus", default=None, help="device list '0,1,2,3' or a count (reference train_transformer.py:39,46); more than "
                                                 "one device and no torch.distributed environment: the ranks are started from here")
    ap.add_argument("--check_val_every_n_epoch", type=int, default=1, help="Lightning's flag (scripts/train_config_log_gpus.sh:3)")
    ap.add_argument("--limit_val_batches", type=int, default=0, help="0 = the whole validation set")
    args, unknown = ap.parse_known_args()

    from .launch import spawn_ranks_if_needed, in_distributed_env
    ngpu = 1
    if not in_distributed_env():          # a rank of the launched job inherits the mask its parent composed: do not apply it twice
        ngpu, ids = parse_gpus(args.gpus)
        if ids is not None:
            os.environ["HIP_VISIBLE_DEVICES"] = ",".join(ids)       # rank r -> r-th listed device
    import sys
    rc = spawn_ranks_if_needed(ngpu, "mebt_amd.train", sys.argv[1:], module=True)      # before anything touches the GPU
    if rc is not None:
        sys.exit(rc)

    import torch.distributed as dist
    from . import presets
    from .config import load_config
    from .parallel import GradReducer
    from .trainer import TrainLoop

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from .parallel import init_rccl
        init_rccl(rank, world, dev)
    random.seed(42); np.random.seed(42); torch.manual_seed(42)          # pl.seed_everything(42)

    cfg = getattr(presets, args.preset)() if args.preset else load_config(args.base, unknown)
    cfg.model.params.class_cond_dim = None
    model = presets.build_model(cfg, compute_dtype=args.dtype)
    ckpt = None
    if args.ckpt_path:
        from .lightning_shim import load_checkpoint_file      # reference checkpoints pickle OmegaConf / Lightning objects
        ckpt = load_checkpoint_file(args.ckpt_path)
        model.load_state_dict(ckpt["state_dict"], strict=False)
    model = model.to(dev).train()
    accum = cfg.exp.get("accumulate_grad_batches", None) or args.accumulate_grad_batches or 1      # train_transformer.py:46-49
    loop = TrainLoop(model, GradReducer(world_size=world), max_steps=args.max_steps, accumulate_grad_batches=accum)
    start_step = 0
    if ckpt is not None and "mebt_amd_loop" in ckpt:      # resume: optimizer moments, step counters, RNG (like trainer.fit(ckpt_path=))
        loop.load_state_dict(ckpt["mebt_amd_loop"])
        start_step = int(ckpt.get("global_step", 0))

    shape = list(cfg.model.mask.params.shape)
    # data: the reference's `vtokens` contract (mebt/data.py:236-305,330-414) — token clips from an .npz / .h5 file with
    # {train,test}_data / _idx, or synthetic grids; every item carries indices = randperm(T*H*W); ranks read disjoint shards
    from .config import AttrDict
    from .data import TokenData
    dargs = AttrDict(dict(cfg.data)) if "data" in cfg else AttrDict()
    if args.tokens:
        dargs["data_path"] = args.tokens
    # token files are in latent units: clip length / crop come from the mask grid (the config's data node of the shipped
    # YAMLs is in pixel units, e.g. sequence_length 16 frames -> 4 latent frames), the file's own grid is its resolution
    dargs["latent_shape"] = shape
    dargs["sequence_length"] = shape[0] * int(dargs.get("sample_every_n_frames", 1))
    dargs["resolution"] = None
    dargs["spatial_length"] = shape[1]
    dargs.setdefault("batch_size", 6)
    data = TokenData(dargs, world_size=world, rank=rank)
    loader = data.train_dataloader()
    # `opt_step` counts OPTIMIZER steps (Lightning's global_step: --max_steps, the checkpointed 'global_step' and the LR /
    # beta(t) / t_prior schedules all count those); with accumulate_grad_batches = k a step is k micro-batches
    per_epoch = max(1, len(loader))
    micro = int(ckpt["mebt_amd_loop"].get("micro_batches", start_step * accum)) if (ckpt is not None and "mebt_amd_loop" in ckpt) else 0
    epoch, skip = micro // per_epoch, micro % per_epoch
    if hasattr(loader.sampler, "set_epoch"):
        loader.sampler.set_epoch(epoch)
    it = iter(loader)
    for _ in range(skip):                 # resume inside an epoch: the batches already consumed are not replayed
        next(it)

    def validate():
        """Lightning's validation loop around validation_step (reference transformer.py:741-747): eval mode, no gradients,
        one `t` per batch from the python RNG as in training; mean loss / top-1 / top-5 over the set, averaged over ranks."""
        vl = data.val_dataloader()
        model.eval()
        tot = torch.zeros(4, dtype=torch.float64, device=dev)
        for bi, vb in enumerate(vl):
            if args.limit_val_batches and bi >= args.limit_val_batches:
                break
            vb = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in vb.items()}
            with torch.no_grad():
                acc1, acc5, loss, _ = model.shared_step(vb, bi)
            tot += torch.stack([loss.double().reshape(()), torch.as_tensor(acc1, device=dev).double().reshape(()),
                                torch.as_tensor(acc5, device=dev).double().reshape(()), torch.ones((), dtype=torch.float64, device=dev)])
        model.train()
        tot = loop.reducer.mean_scalars(tot).cpu()
        n = max(1.0, float(tot[3]))
        return float(tot[0]) / n, float(tot[1]) / n, float(tot[2]) / n, int(tot[3])

    t0 = time.perf_counter()
    opt_step = start_step

    def fetch():
        try:
            return next(it)
        except StopIteration:
            return None

    batch = fetch()
    while opt_step < args.max_steps:
        if batch is None:
            epoch += 1
            if args.check_val_every_n_epoch and epoch % args.check_val_every_n_epoch == 0:
                model.current_epoch = epoch - 1            # Lightning counts the epoch that just ended from 0
                if rank == 0:                              # sample -> decode -> logger video every vis_epoch epochs (transformer.py:336-351)
                    if model.logger is None and model.first_stage_model is not None:
                        from .lightning_shim import VideoLogger
                        model.logger = VideoLogger(os.path.join(args.default_root_dir, "videos"))
                    rng = (random.getstate(), np.random.get_state(), torch.get_rng_state())     # the visualisation draws on rank 0 only:
                    model.on_validation_epoch_start()                                            # keep the ranks' host RNG streams aligned
                    random.setstate(rng[0]); np.random.set_state(rng[1]); torch.set_rng_state(rng[2])
                vloss, v1, v5, nb = validate()
                if rank == 0:
                    print(f"epoch {epoch}: val/loss {vloss:.4f} acc1 {v1:.2f} acc5 {v5:.2f} ({nb} batches)", flush=True)
            if hasattr(loader.sampler, "set_epoch"):
                loader.sampler.set_epoch(epoch)
            it = iter(loader)
            batch = fetch()
        x, idx = batch["video"], batch["indices"]
        batch = fetch()                   # look ahead: the last batch of an epoch closes its accumulation group (Lightning steps
        before = loop.step_count          # the optimizer on is_last_batch), so no group spans the validation pass
        stats = loop.step(x.to(dev, non_blocking=True), idx.to(dev, non_blocking=True), flush=batch is None)
        micro += 1
        if loop.step_count == before:     # a non-final micro-batch of an accumulation group
            continue
        opt_step += 1
        if opt_step % args.log_every == 0:
            s = loop.reducer.mean_scalars(torch.stack([stats[4], 100 * stats[1] / stats[3], 100 * stats[2] / stats[3]])).cpu()
            if rank == 0:
                dt = time.perf_counter() - t0
                print(f"step {opt_step}: train/loss {s[0]:.4f} acc1 {s[1]:.2f} acc5 {s[2]:.2f} "
                      f"lr {model.learning_rate * model.lr_scale():.3e}  {dt / (opt_step - start_step) * 1e3:.1f} ms/step", flush=True)
        if args.ckpt_every and opt_step % args.ckpt_every == 0:
            loop.consolidate()           # sharded optimizer: a collective on all ranks, then rank 0 writes
        if args.ckpt_every and rank == 0 and opt_step % args.ckpt_every == 0:
            os.makedirs(args.default_root_dir, exist_ok=True)
            lsd = loop.state_dict()
            lsd["micro_batches"] = micro
            torch.save({"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                        "hyper_parameters": model.hparams, "global_step": opt_step, "epoch": epoch,
                        "mebt_amd_loop": lsd},
                       os.path.join(args.default_root_dir, f"step={opt_step}.ckpt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
