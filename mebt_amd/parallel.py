"""Single-node data parallelism: one process per GPU, full replica, gradient all-reduce over
RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm).  Counterpart of the reference's only
multi-GPU mechanism, Lightning's DDPStrategy (train_transformer.py:39-41).

Design: gradients live in two flat fp32 buffers laid out in layer order, so a
bucket is a contiguous slice — no flatten/unflatten copies.  The HIP backward is split at bucket
boundaries (mebt_backward_head / _layers / _embed); after each piece the reducer launches an
asynchronous all-reduce of the slice that just became final.  RCCL runs it on its own stream after
an event on the compute stream, so the collective overlaps the rest of backward; only the last
bucket (P: LN/bias/embedding gradients, 19 M elements at C2) is exposed.  Gradients are summed; the
1/world_size factor is folded into the fused AdamW kernel (`grad_scale`).  All ranks draw the same
`t` (same python seed, train_transformer.py:11), hence identical NC/NT and no stragglers.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, world_size=None, group=None, layers_per_bucket=4):
        self.group = group
        self.world_size = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.grad_scale = 1.0 / self.world_size
        self.layers_per_bucket = layers_per_bucket
        self.pending = []
        self.bytes_reduced = 0

    # called right after the kernels producing a bucket have been enqueued on the current stream
    def bucket_ready(self, native, stage, hi, lo):
        """Launch the asynchronous all-reduce(s) of the gradient slices that `stage` finalised; returns the
        list of work handles (empty when world_size == 1)."""
        if self.world_size == 1:
            return []
        if stage == "head":
            a, b = native.head_w_range()
            views = [native.gW[a:b]]
        elif stage == "layers":
            a, b = native.layer_w_range(hi, lo)
            c, d = native.layer_p_range(hi, lo)
            views = [native.gW[a:b], native.gP[c:d]]
        else:                       # 'embed': ln_f + embedding gradients are final
            c, d = native.tail_p_range()
            views = [native.gP[c:d]]
        works = []
        for v in views:
            works.append(dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.bytes_reduced += v.numel() * v.element_size()
        self.pending += works
        return works

    def wait(self):
        """make the compute stream wait for every outstanding bucket (no host synchronisation on GPU)"""
        for w in self.pending:
            w.wait()
        self.pending = []

    def broadcast_parameters(self, native, src=0):
        """DDP's start-up broadcast rank0 -> all"""
        if self.world_size == 1:
            return
        dist.broadcast(native.W, src=src, group=self.group)
        dist.broadcast(native.P, src=src, group=self.group)
        if getattr(native, "Wlp", None) is not None:
            native.sync_lowp(force=True)

    def mean_scalars(self, t):
        """one small all-reduce for the logged scalars (the reference issues four: loss, acc1, acc5, lr)"""
        if self.world_size == 1:
            return t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world_size
