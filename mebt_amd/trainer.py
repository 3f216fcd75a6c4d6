"""The training step as the launcher and bench.py drive it: forward -> fused loss -> backward with
bucketed all-reduce overlap -> fused AdamW, all on the engine, no host synchronisation inside.
Counterpart of what Lightning's fit loop does around the reference's training_step / optimizer_step
(SURVEY.md §3.1)."""
import os
import random

import numpy as np
import torch

from .parallel import GradReducer, pick_concurrent_stream


class TrainLoop:
    def __init__(self, model, reducer=None, max_steps=0, overlap_optimizer=None, fused_optimizer=None,
                 accumulate_grad_batches=1):
        if overlap_optimizer is None:   # pays when there is an all-reduce to hide behind; on one GPU both contend for HBM
            multi = reducer is not None and getattr(reducer, "active", reducer.world_size > 1)
            overlap_optimizer = os.environ.get("MEBT_OVERLAP_OPT", "1" if multi else "0") != "0"
        """model: mebt_amd.transformer.Net2NetTransformer on its GPU, with learning_rate /
        warmup_steps / weight_decay / cosine_lr set (train_transformer.py:54-66)."""
        self.model = model
        self.accum = max(1, int(accumulate_grad_batches))
        self._micro = 0
        self.native = model._ensure_native()
        self.native.ensure_grads()
        self.reducer = reducer or GradReducer(world_size=1)
        model._reducer = self.reducer
        self.step_count = 0
        model.trainer.max_steps = max_steps
        self.reducer.broadcast_parameters(self.native)
        # the optimizer runs bucket by bucket on its own stream, as soon as a bucket's gradients are final
        # (after its all-reduce when data-parallel): AdamW streams 30 B/parameter through HBM while the rest
        # of backward is latency/compute-bound, so the two overlap almost perfectly
        # ... provided the two streams do not share a hardware queue: probed, not assumed (parallel.pick_concurrent_stream)
        self.opt_stream = pick_concurrent_stream(torch.cuda.current_stream(), device=self.native.device) if overlap_optimizer else None
        # The engine's bf16 wire-gradient binding must match the step() branch that will run: with it bound the Linear weight
        # gradients exist ONLY in gWb (bf16), which only reduce_update() reads.  Set it either way — a TrainLoop built on a
        # model whose previous loop had bound it (bench.py's fallback) must not inherit the binding.
        # data parallel: one tuner (rank 0), every rank launches its choices (parallel.GradReducer.sync_tune_table)
        self.sync_tune = bool(self.reducer.active and os.environ.get("MEBT_DP_SYNC_TUNE", "1") != "0")
        if self.sync_tune:
            self.reducer.lead_tuning()
        # a benchmark sets this around its timed region (after one explicit sync_tune_table() behind its warm-up): the sync
        # points are host reads of a broadcast, i.e. they drain the GPU queue of every rank
        self.hold_tune_sync = False
        self.sharded = bool(self.reducer.active and self.reducer.mode == "sharded")
        wire = (self.sharded and self.reducer.wire == "bf16" and self.accum == 1 and model.compute_dtype == "bf16"
                and os.environ.get("MEBT_DP_WIRE_GRADS", "1") != "0")
        if self.sharded:
            self.native._adam_state()           # allocated up front: the first sharded update runs on the optimizer stream
        self.wire_grads = bool(wire)
        self.native.enable_wire_grads(wire)     # weight gradients leave the MFMA epilogue in the wire format (or not)
        # one process, bf16: AdamW of the blocks' Linear weights is applied inside the weight-gradient launches of
        # backward (the gradient never goes to HBM and the optimizer traffic hides behind MFMA work); the all-reduce of a
        # data-parallel job needs the gradients first, so this is the single-GPU path only
        if fused_optimizer is None:   # the fp32 parity mode has no fused epilogue (the engine would fall back to one AdamW launch per weight)
            fused_optimizer = os.environ.get("MEBT_FUSED_ADAMW", "1") != "0" and model.compute_dtype == "bf16"
        self.fused_optimizer = bool(fused_optimizer) and not self.reducer.active
        # gradient accumulation (train_transformer.py:46-49 -> Lightning accumulate_grad_batches): `step` is called once per
        # micro-batch; gradients of k consecutive calls are averaged (loss / k, as Lightning scales it), the all-reduce, the
        # optimizer and the step counters run on the k-th.  The optimizer-in-backward needs the whole gradient in one
        # backward, so accumulation uses the separate optimizer.
        if self.accum > 1:
            self.fused_optimizer = False

    def step(self, x, indices, t=None, flush=False):
        """x [B,T,H,W] int64 tokens, indices [B,N] permutations.  Returns a device tensor
        [CE sum, #top1, #top5, #rows, loss] (float64) — read it only when you want to log.
        flush (gradient accumulation only): this is the last batch of the epoch — run the optimizer now even if the group
        is short, as Lightning does (its accumulation scheduler steps on `is_last_batch`; the loss stays divided by k)."""
        m, nm, red = self.model, self.native, self.reducer
        B = x.shape[0]
        x_ids = x.reshape(B, -1)
        if t is None:
            t = m._draw_t(False)
        prior_t = m.t_prior(m.t_lengths, m.global_step)
        ci, ti, seq_len = m.mask_sampler.divide_indices(indices, torch.tensor(float(t)), m.t_lengths, prior_t)
        ci, ti = ci.contiguous(), ti.contiguous()
        ratio = float(seq_len - ci.shape[1]) / float(seq_len)
        scale = 1.0 / (B * seq_len * ratio ** m.config.avg_loss)
        lr = m.learning_rate * m.lr_scale()
        logits = nm.forward(x_ids, ci, ti, training=True, dropout_seed=m._next_seed())
        # loss statistics and d(loss)/d(logits) from one pass over the logits (the scale every backward call below uses)
        fused_ce = os.environ.get("MEBT_FUSED_CE", "1") != "0"      # 0: separate cross-entropy forward / backward kernels (A/B)
        stats = nm.loss_stats(logits, grad_scale=(scale / self.accum if self.accum > 1 else scale) if fused_ce else None)
        if self.accum > 1:
            self._micro += 1
            nm.set_grad_accumulate(self._micro > 1)
            if self._micro < self.accum and not flush:     # not the last micro-batch: local accumulation only (DDP no_sync)
                nm.backward(logits, scale / self.accum)
                nm.set_grad_accumulate(False)
                return torch.cat([stats, (stats[0] * scale).reshape(1)])
            self._micro = 0
            bwd_scale = scale / self.accum
        else:
            bwd_scale = scale
        self.step_count += 1
        if self.fused_optimizer:
            nm.set_fused_adamw(lr, m.weight_decay, self.step_count)
            nm.backward(logits, scale)
            nm.set_fused_adamw(step=0)
            nm.adamw_range("rest", None, None, lr, m.weight_decay, self.step_count)
        elif self.opt_stream is None and not self.sharded:
            nm.backward(logits, bwd_scale, between=lambda s, hi, lo: red.bucket_ready(nm, s, hi, lo))
            red.wait()
            nm.adamw_step(lr, m.weight_decay, self.step_count, grad_scale=red.grad_scale)
        elif self.sharded:
            # reduce-scatter -> AdamW on this rank's shard -> all-gather, bucket by bucket behind the backward (parallel.py);
            # without an optimizer stream (MEBT_OVERLAP_OPT=0) the same calls run in line on the compute stream
            opt = self.opt_stream
            update = lambda s, hi, lo: red.reduce_update(nm, s, hi, lo, lr, m.weight_decay, self.step_count, opt_stream=opt)
            if self.step_count == 1:
                # the first backward is where the GEMM tuner times its candidates (in situ, on this stream): keep RCCL's
                # kernels and the shard updates out of the way until it is through, or a noisy pick on one rank makes that
                # rank the straggler of every later step
                done = []
                nm.backward(logits, bwd_scale, bucket_layers=red.bucket_plan(nm.n_layer), between=lambda s, hi, lo: done.append((s, hi, lo)))
                for stage in done:
                    update(*stage)
            else:
                nm.backward(logits, bwd_scale, bucket_layers=red.bucket_plan(nm.n_layer), between=update)
            if red.defer:
                red.gather_deferred(nm, opt)                        # the next forward waits for the gathers bucket by bucket
            else:
                red.finish(opt)                                     # the next forward reads the gathered weights
        else:
            main, opt = torch.cuda.current_stream(), self.opt_stream

            def bucket(stage, hi, lo):
                works = red.bucket_ready(nm, stage, hi, lo)         # async all-reduce of this bucket's slices ([] when N = 1)
                opt.wait_stream(main)                               # the bucket's gradients are enqueued on `main`
                with torch.cuda.stream(opt):
                    for w in works:
                        w.wait()                                    # ... and reduced across ranks
                    nm.adamw_range(stage, hi, lo, lr, m.weight_decay, self.step_count, grad_scale=red.grad_scale)

            nm.backward(logits, bwd_scale, between=bucket)
            red.pending = []
            main.wait_stream(opt)                                   # the next forward reads the updated weights
        if self.accum > 1:
            nm.set_grad_accumulate(False)
        if self.sync_tune and not self.hold_tune_sync and red.tune_sync_due(self.step_count):
            red.sync_tune_table()       # rank 0 is the only rank that tunes GEMM tiles in situ: the others adopt its table
        m.trainer.global_step += 1
        m.global_step += 1
        return torch.cat([stats, (stats[0] * scale).reshape(1)])

    def consolidate(self, optimizer_state=True):
        """Data-parallel sharded optimizer: all-gather the fp32 masters (and moments) so that every rank — in particular
        the one that writes the checkpoint — holds complete tensors.  A collective: call on all ranks."""
        self.reducer.consolidate(self.native, optimizer_state=optimizer_state)

    # ---- resume (reference: trainer.fit(..., ckpt_path=...) restores optimizer, step counters and RNG) ---------------
    def state_dict(self):
        """Everything a resumed run needs besides the model's state_dict: AdamW moments, the optimizer step (bias
        correction), the step counters that drive the LR / beta(t) / t_prior schedules and the dropout seeds, and the
        python / numpy / torch RNG states (`t` is drawn from the python RNG, mebt/transformer.py:228)."""
        m = self.model
        return {"adam": [t.detach().cpu().clone() for t in self.native._adam_state()], "step_count": self.step_count,
                "global_step": m.global_step, "trainer_global_step": m.trainer.global_step, "seed_ctr": m._seed_ctr,
                "rng": {"python": random.getstate(), "numpy": np.random.get_state(), "torch": torch.get_rng_state()}}

    def load_state_dict(self, sd, restore_rng=True):
        m = self.model
        for dst, src in zip(self.native._adam_state(), sd["adam"]):
            dst.copy_(src)
        self.step_count = int(sd["step_count"])
        m.global_step = int(sd["global_step"])
        m.trainer.global_step = int(sd["trainer_global_step"])
        m._seed_ctr = int(sd["seed_ctr"])
        if restore_rng and "rng" in sd:
            random.setstate(sd["rng"]["python"])
            np.random.set_state(sd["rng"]["numpy"])
            torch.set_rng_state(sd["rng"]["torch"])
