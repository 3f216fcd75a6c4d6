"""The training step as the launcher and bench.py drive it: forward -> fused loss -> backward with
bucketed all-reduce overlap -> fused AdamW, all on the engine, no host synchronisation inside.
Counterpart of what Lightning's fit loop does around the reference's training_step / optimizer_step
(SURVEY.md §3.1)."""
import os
import random

import numpy as np
import torch

from .parallel import GradReducer


class TrainLoop:
    def __init__(self, model, reducer=None, max_steps=0, overlap_optimizer=None, fused_optimizer=None):
        if overlap_optimizer is None:   # pays when there is an all-reduce to hide behind; on one GPU both contend for HBM
            world = reducer.world_size if reducer is not None else 1
            overlap_optimizer = os.environ.get("MEBT_OVERLAP_OPT", "1" if world > 1 else "0") != "0"
        """model: mebt_amd.transformer.Net2NetTransformer on its GPU, with learning_rate /
        warmup_steps / weight_decay / cosine_lr set (train_transformer.py:54-66)."""
        self.model = model
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
        self.opt_stream = torch.cuda.Stream(device=self.native.device) if overlap_optimizer else None
        # one process, bf16: AdamW of the blocks' Linear weights is applied inside the weight-gradient launches of
        # backward (the gradient never goes to HBM and the optimizer traffic hides behind MFMA work); the all-reduce of a
        # data-parallel job needs the gradients first, so this is the single-GPU path only
        if fused_optimizer is None:   # the fp32 parity mode has no fused epilogue (the engine would fall back to one AdamW launch per weight)
            fused_optimizer = os.environ.get("MEBT_FUSED_ADAMW", "1") != "0" and model.compute_dtype == "bf16"
        self.fused_optimizer = bool(fused_optimizer) and self.reducer.world_size == 1

    def step(self, x, indices, t=None):
        """x [B,T,H,W] int64 tokens, indices [B,N] permutations.  Returns a device tensor
        [CE sum, #top1, #top5, #rows, loss] (float64) — read it only when you want to log."""
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
        self.step_count += 1
        logits = nm.forward(x_ids, ci, ti, training=True, dropout_seed=m._next_seed())
        stats = nm.loss_stats(logits)
        if self.fused_optimizer:
            nm.set_fused_adamw(lr, m.weight_decay, self.step_count)
            nm.backward(logits, scale)
            nm.set_fused_adamw(step=0)
            nm.adamw_range("rest", None, None, lr, m.weight_decay, self.step_count)
        elif self.opt_stream is None:
            nm.backward(logits, scale, between=lambda s, hi, lo: red.bucket_ready(nm, s, hi, lo))
            red.wait()
            nm.adamw_step(lr, m.weight_decay, self.step_count, grad_scale=red.grad_scale)
        else:
            main, opt = torch.cuda.current_stream(), self.opt_stream

            def bucket(stage, hi, lo):
                works = red.bucket_ready(nm, stage, hi, lo)         # async all-reduce of this bucket's slices ([] when N = 1)
                opt.wait_stream(main)                               # the bucket's gradients are enqueued on `main`
                with torch.cuda.stream(opt):
                    for w in works:
                        w.wait()                                    # ... and reduced across ranks
                    nm.adamw_range(stage, hi, lo, lr, m.weight_decay, self.step_count, grad_scale=red.grad_scale)

            nm.backward(logits, scale, between=bucket)
            red.pending = []
            main.wait_stream(opt)                                   # the next forward reads the updated weights
        m.trainer.global_step += 1
        m.global_step += 1
        return torch.cat([stats, (stats[0] * scale).reshape(1)])
