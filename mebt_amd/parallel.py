"""Single-node data parallelism: one process per GPU, full replica of the network, RCCL over xGMI
(torch.distributed backend "nccl" is RCCL on ROCm).  Counterpart of the reference's only multi-GPU mechanism,
Lightning's DDPStrategy: all-reduce(mean) of 337 M gradients + a replicated optimizer (train_transformer.py:39-41).

Design.  Gradients live in two flat fp32 buffers laid out in layer order, so a bucket is a
contiguous slice — no flatten / unflatten copies.  The HIP backward is split at bucket boundaries
(mebt_backward_head / _layers / _embed); as soon as a bucket's gradients are final the reducer runs, per bucket,

  mode "sharded" (default):   bf16 wire buffer                         (Linear weight gradients: stored as bf16 by the
                                                                        weight-gradient launches themselves; the small
                                                                        non-Linear tail: cast kernel on the compute stream)
                              all-to-all of the bucket's shards        (RCCL, its own stream)     -> rank r holds every rank's
                                                                        bf16 copy of shard r; summed in fp32 inside the AdamW
                                                                        launch (mebt_adamw_slice_pieces).  MEBT_DP_EXCHANGE=rs:
                                                                        reduce-scatter(sum), RCCL adds in bf16
                              AdamW on shard r only                    (mebt_adamw_slice[_pieces], optimizer stream; fp32 master,
                                                                        m, v are touched by the owner alone: 1/N of the
                                                                        10 GB the replicated optimizer streams per step)
                              all-gather of the updated shard          (RCCL): the bf16 weight mirror for the Linear
                                                                        weights, fp32 for the small non-Linear tail
  mode "allreduce" (legacy):  all-reduce(sum) of the fp32 bucket, replicated bucket-wise AdamW (mebt_adamw_range)

Everything after the cast overlaps the rest of backward; only the last bucket is exposed.  On the wire a step moves
2 x 2 B per parameter (all-to-all / reduce-scatter + all-gather in bf16) instead of 2 x 4 B for the fp32 all-reduce; each rank's
bucket was accumulated in fp32 locally and rounded once, the sum across ranks is taken in fp32 (all-to-all; in bf16 by RCCL with
MEBT_DP_EXCHANGE=rs).  The
1/world_size factor is folded into AdamW (`grad_scale`).  All ranks draw the same `t` (same python seed,
train_transformer.py:11), hence identical NC/NT and no stragglers.

In sharded bf16 mode the fp32 master copy of W is up to date only on the owning rank between steps (the compute path
reads the bf16 mirror, which IS complete): `consolidate()` all-gathers the masters (and optionally the optimizer
moments) before a checkpoint / state_dict.
"""
import os

import torch
import torch.distributed as dist


def init_rccl(rank, world_size, device):
    """torch.distributed over RCCL with RCCL's own stream taken from the HIGH-priority pool: on MI355X high- and low-priority
    HIP streams never share a hardware queue (streams_serialised below), so the collectives can neither be stuck behind the
    compute stream's kernels nor hold them back, whatever queue the low-priority streams of this process landed on."""
    try:
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    except Exception:                # a torch build without the option: default stream priority, everything else unchanged
        opts = None
    if opts is not None:
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=device, pg_options=opts)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=device)


def streams_serialised(a, b, cycles=4_000_000):
    """Do kernels on stream `b` queue up behind kernels on stream `a`?  HIP maps streams onto a few hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default; measured on MI355X: stream k of torch's pool shares a queue with streams k +- 4, and the
    default stream with every fourth pool stream; high- and low-priority streams never share one).  Two streams on one
    queue run in submission order — including the event waits in it, so "overlap" between them is none at all.  Measured,
    not assumed: a ~2 ms spin kernel on `a`, then a trivial kernel on `b`; if `a`'s spin is over when `b`'s kernel
    completes, `b` waited."""
    x = torch.zeros(64, device=torch.device("cuda", torch.cuda.current_device()))
    torch.cuda.synchronize()
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
        ea.record()
    with torch.cuda.stream(b):
        x.add_(1)
        eb.record()
    eb.synchronize()
    waited = ea.query()
    torch.cuda.synchronize()
    return bool(waited)


def pick_concurrent_stream(ref, device=None, tries=12, priority=0):
    """a new stream whose kernels really run beside those of `ref` (see streams_serialised)"""
    s = None
    streams_serialised(ref, ref)                          # warm-up: first launches, lazy stream creation
    for _ in range(tries):
        s = torch.cuda.Stream(device=device, priority=priority)
        if not streams_serialised(s, ref) and not streams_serialised(ref, s):
            return s
    return s


class _Done:
    """work handle of a collective that already completed (host-staged functional path)"""

    def wait(self):
        return True


class GradReducer:
    def __init__(self, world_size=None, group=None, layers_per_bucket=4, mode=None, wire=None, force=None, exchange=None):
        self.group = group
        self.world_size = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = dist.get_rank(group) if (self.world_size > 1 and dist.is_initialized()) else 0
        # `force`: run the collective path even with one rank (an initialised process group of size 1): every RCCL call, dtype,
        # buffer aliasing and stream hand-off of the N > 1 path executes on a single GPU (tests/test_gpu_dp.py)
        force = (os.environ.get("MEBT_DP_FORCE", "0") == "1") if force is None else force
        self.active = self.world_size > 1 or (bool(force) and dist.is_initialized())
        self.grad_scale = 1.0 / self.world_size
        self.layers_per_bucket = layers_per_bucket
        self.taper = os.environ.get("MEBT_DP_TAPER", "1") != "0"
        # deferred gathers: the all-gathers of the updated shards are issued after the whole backward, in the order the NEXT
        # forward reads the parameters, and that forward waits bucket by bucket (mebt_model_set_forward_waits) instead of
        # waiting for all of them up front: the reduce-scatters get the backward window to themselves and the gathers
        # overlap the forward
        self.defer = os.environ.get("MEBT_DP_DEFER_GATHER", "1") != "0"
        self._deferred = []          # (first layer that reads it, full, mine) per sharded bucket
        self._repl_keys = []         # first-reading layers of buckets updated the replicated way (no gather, but the forward still waits for their AdamW)
        self._last_opt = self._native = None
        self._test_delay = int(os.environ.get("MEBT_DP_TEST_DELAY_CYCLES", "0"))
        self.mode = mode or os.environ.get("MEBT_DP_MODE", "sharded")
        self.wire = wire or os.environ.get("MEBT_DP_WIRE", "bf16")
        # how the bf16 buckets are summed across ranks: "a2a" = all-to-all of the shards, the N received copies added in fp32 inside
        # the owner's AdamW launch (the reference's DDP sums fp32, train_transformer.py:39-41; on xGMI's full mesh every link
        # carries one shard at once); "rs" = RCCL reduce-scatter, which adds in bf16 (one rounding per hop).  fp32 wire: always "rs".
        self.exchange = exchange or os.environ.get("MEBT_DP_EXCHANGE", "a2a")
        assert self.mode in ("sharded", "allreduce") and self.wire in ("bf16", "fp32") and self.exchange in ("a2a", "rs")
        # `elide` (bench.py, measurement only): every collective of the step becomes a local no-op / copy with the same
        # buffers, kernels and stream hand-offs — the step time with it set is the compute side of the data-parallel step,
        # the difference to the real step is communication that was not hidden (plus RCCL's kernels competing for CUs)
        self.elide = False
        self.pending = []            # outstanding collectives the compute stream has to wait for
        self.bytes_on_wire = 0       # payload bytes handed to collectives (per rank, per direction)
        self.master_stale = False    # sharded + bf16 mirror: fp32 W is current only on the owner of each shard
        self._wire_buf = {}
        self._rs_out = {}
        self._sharded_ranges = set()     # (which, start, end) of every bucket range that was cut into shards: what consolidate() gathers
        self._inplace = not (self.active and dist.is_initialized() and dist.get_backend(group) != "nccl")

    # ---- collectives -------------------------------------------------------------------------------------------------------
    # RCCL: asynchronous, ordered after the current stream, in place where send and receive buffers alias.  A gloo group
    # driving GPU tensors (functional runs of the N > 1 path with all ranks on one GPU: tests, MEBT_BENCH_SHARE_GPU) is
    # staged through host memory synchronously — same arithmetic, no overlap.
    def _staged(self, t):
        return (not self._inplace) and t.is_cuda

    def _reduce_scatter(self, out, inp):
        if self.elide:               # measurement only: this rank's own contribution stands in for the sum
            n = out.numel()
            out.copy_(inp[self.rank * n:(self.rank + 1) * n])
            return _Done()
        if self._staged(inp):
            ho = torch.empty(out.shape, dtype=out.dtype)
            dist.reduce_scatter_tensor(ho, inp.cpu(), op=dist.ReduceOp.SUM, group=self.group)
            out.copy_(ho)
            return _Done()
        return dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _all_to_all(self, recv, send):
        """recv[j * shard : (j + 1) * shard] = rank j's send[rank * shard : (rank + 1) * shard]"""
        if self.elide:               # measurement only: N copies of this rank's own shard stand in for the ranks' contributions
            n = recv.numel() // self.world_size
            recv.view(self.world_size, n).copy_(send[self.rank * n:(self.rank + 1) * n].unsqueeze(0).expand(self.world_size, n))
            return _Done()
        if self._staged(send):
            hr = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(hr, send.cpu(), group=self.group)
            recv.copy_(hr)
            return _Done()
        return dist.all_to_all_single(recv, send, group=self.group, async_op=True)

    def _all_gather(self, full, mine):
        if self.elide:
            return _Done()
        if self._staged(full):
            hf = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(hf, mine.cpu(), group=self.group)
            full.copy_(hf)
            return _Done()
        return dist.all_gather_into_tensor(full, mine if self._inplace else mine.clone(), group=self.group, async_op=True)

    # ---- bucket geometry ------------------------------------------------------------------------------------------------
    def bucket_plan(self, n_layer):
        """Layer counts of the gradient buckets in the order backward finishes them (top of the network first).  Uniform
        `layers_per_bucket` groups, except that the last buckets taper (... 4, 2, 1, 1): whatever follows the final
        layer's gradients — reduce-scatter, shard AdamW, all-gather — is not hidden behind any backward work, so the
        last bucket should be the smallest one (a 1-layer bucket is 25 MB of bf16 per direction instead of 100 MB)."""
        plan, rem = [], int(n_layer)
        while rem > 0:
            size = min(self.layers_per_bucket, max(1, rem // 2)) if self.taper else min(self.layers_per_bucket, rem)
            plan.append(size)
            rem -= size
        return plan

    @staticmethod
    def bucket_ranges(native, stage, hi, lo):
        """[(which, start, end)] in the flat W (which = 0) / P (which = 1) buffers that `stage` finalised"""
        if stage == "head":
            a, b = native.head_w_range()
            return [(0, a, b)]
        if stage == "layers":
            a, b = native.layer_w_range(hi, lo)
            c, d = native.layer_p_range(hi, lo)
            return [(0, a, b), (1, c, d)]
        c, d = native.tail_p_range()            # 'embed': ln_f + embedding gradients are final
        return [(1, c, d)]

    @staticmethod
    def sharded_ranges(native, stage, hi, lo):
        """Buckets of the sharded mode: the Linear weights as above, but ALL non-Linear parameters (the per-layer bias /
        LayerNorm slices, 13 d each, and the embedding tail) as one bucket at the end — the embedding gradients, 90 % of
        it, are final only then anyway, and eight pairs of latency-bound 50 k-element collectives per step would otherwise
        sit in the RCCL queue between the large ones."""
        if stage == "head":
            return [(0,) + tuple(native.head_w_range())]
        if stage == "layers":
            return [(0,) + tuple(native.layer_w_range(hi, lo))]
        return [(1, 0, native.gP.numel())]

    # ---- legacy: bucketed fp32 all-reduce ---------------------------------------------------------------------------------
    def bucket_ready(self, native, stage, hi, lo):
        """Launch the asynchronous all-reduce(s) of the gradient slices that `stage` finalised; returns the
        list of work handles (empty when world_size == 1)."""
        if not self.active:
            return []
        works = []
        for which, a, b in self.bucket_ranges(native, stage, hi, lo):
            v = (native.gW if which == 0 else native.gP)[a:b]
            if self.elide:
                continue
            works.append(dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.bytes_on_wire += v.numel() * v.element_size()
        self.pending += works
        return works

    # ---- sharded: reduce-scatter -> shard AdamW -> all-gather ----------------------------------------------------------------
    def _buf(self, table, key, n, dtype, device):
        t = table.get(key)
        if t is None or t.numel() != n or t.dtype != dtype:
            t = table[key] = torch.empty(n, dtype=dtype, device=device)
        return t

    def reduce_update(self, native, stage, hi, lo, lr, weight_decay, step, opt_stream=None, betas=(0.9, 0.95), eps=1e-8):
        """One gradient bucket, start to finish (see the module docstring).  Call on the compute stream right after the
        bucket's backward was enqueued; the optimizer and the all-gather run on `opt_stream`."""
        world, rank = self.world_size, self.rank
        gpu = native.gW.is_cuda
        main = torch.cuda.current_stream() if gpu else None
        self._last_opt, self._native = opt_stream, native
        # the first thing the next forward does with this bucket: -1 = its first kernel (embeddings, biases, LayerNorms),
        # a block index, or n_layer = the head
        key = native.n_layer if stage == "head" else (lo if stage == "layers" else -1)
        for which, a, b in self.sharded_ranges(native, stage, hi, lo):
            n = b - a
            if n == 0:
                continue
            g = (native.gW if which == 0 else native.gP)[a:b]
            gWb = getattr(native, "gWb", None) if which == 0 else None      # bound: the Linear weight gradients exist ONLY there (bf16)
            if n % (4 * world):      # cannot be cut into aligned shards: replicate (all-reduce + full-range update)
                gsrc = gWb[a:b] if gWb is not None else g
                if self.elide:
                    w = _Done()
                elif self._staged(gsrc):
                    h = gsrc.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                    gsrc.copy_(h)
                    w = _Done()
                else:
                    w = dist.all_reduce(gsrc, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.bytes_on_wire += 2 * n * gsrc.element_size()
                self._on_opt(opt_stream, main, lambda: (w.wait(), native.adamw_slice(which, a, n, gsrc, lr, weight_decay, step, betas=betas,
                                                                                      eps=eps, grad_scale=self.grad_scale,
                                                                                      stream=self._sid(opt_stream))))
                self._repl_keys.append(key)
                continue
            shard = n // world
            sa = a + rank * shard
            self._sharded_ranges.add((which, a, b))
            if self.wire == "bf16" and gWb is not None:
                wire = gWb[a:b]                                                 # the backward already stored bf16 (no fp32 copy, no cast)
            elif self.wire == "bf16":
                wire = self._buf(self._wire_buf, (which, a, b), n, torch.bfloat16, g.device)
                native.cast_bf16(g, wire)                                       # compute stream
            else:
                wire = g
            a2a = self.wire == "bf16" and self.exchange == "a2a"
            pieces = world if a2a else 1
            out = self._buf(self._rs_out, (which, a, b), shard * pieces, wire.dtype, g.device)
            w = self._all_to_all(out, wire) if a2a else self._reduce_scatter(out, wire)
            self.bytes_on_wire += n * wire.element_size()
            lowp = native.Wlp if which == 0 else None

            def tail(w=w, which=which, a=a, b=b, sa=sa, shard=shard, out=out, lowp=lowp, key=key, pieces=pieces):
                w.wait()                                                       # the optimizer stream waits for the exchange
                kw = {"pieces": pieces} if pieces > 1 else {}                  # a2a: the ranks' bf16 contributions are summed in fp32 by the AdamW launch
                native.adamw_slice(which, sa, shard, out, lr, weight_decay, step, betas=betas, eps=eps, grad_scale=self.grad_scale,
                                   stream=self._sid(opt_stream), **kw)
                full_t = lowp if lowp is not None else (native.W if which == 0 else native.P)
                full, mine = full_t[a:b], full_t[sa:sa + shard]
                if self.defer:
                    self._deferred.append((key, full, mine))                   # gather_deferred()
                else:
                    self.pending.append(self._all_gather(full, mine))
                    self.bytes_on_wire += full.numel() * full.element_size()
                if lowp is not None:
                    self.master_stale = True

            self._on_opt(opt_stream, main, tail)

    @staticmethod
    def _sid(stream):
        return stream.cuda_stream if stream is not None else None

    def _on_opt(self, opt_stream, main, fn):
        if opt_stream is None:
            fn()
            return
        opt_stream.wait_stream(main)            # the bucket's gradients (and the cast) are enqueued on the compute stream
        with torch.cuda.stream(opt_stream):
            if self._test_delay:                # tests: make the optimizer / gather side late, so a missing wait shows up as stale weights
                torch.cuda._sleep(self._test_delay)
            fn()

    def gather_deferred(self, native, opt_stream=None):
        """After the last bucket of a step: issue the deferred all-gathers in the order the next forward needs them (non-Linear
        parameters, blocks bottom-up, head) and hand the engine one event per bucket.  The compute stream is NOT made to wait
        here: the next forward — training or inference — waits for each event right before the first block that reads the
        bucket.  Anything else that reads parameters on the compute stream must call finish() first (consolidate() does)."""
        if not self._deferred and not self._repl_keys:
            return
        order, self._deferred = sorted(self._deferred, key=lambda r: r[0]), []
        repl, self._repl_keys = self._repl_keys, []
        gpu = native.gW.is_cuda and hasattr(native, "set_forward_waits")
        main = torch.cuda.current_stream() if native.gW.is_cuda else None
        waits = []

        def run():
            if gpu and repl:                     # their AdamW ran on this stream: one event after all of it
                ev = torch.cuda.Event()
                ev.record()
                waits.extend((k, ev) for k in sorted(set(repl)))
            for key, full, mine in order:
                w = self._all_gather(full, mine)
                self.bytes_on_wire += full.numel() * full.element_size()
                if gpu:
                    w.wait()                     # this (optimizer) stream waits for the gather; RCCL runs them in order anyway
                    ev = torch.cuda.Event()
                    ev.record()
                    waits.append((key, ev))
                else:
                    self.pending.append(w)

        self._on_opt(opt_stream, main, run)
        if gpu:
            native.set_forward_waits(waits)

    def abandon(self, native=None):
        """Give up on this reducer after a failed step: forget half-issued work and undo what it bound on the engine (the
        forward waits and the bf16 wire-gradient binding), so that another reducer / TrainLoop can take the model over."""
        self.pending, self._deferred, self._repl_keys = [], [], []
        native = native if native is not None else self._native
        if native is not None:
            if hasattr(native, "set_forward_waits") and native.gW.is_cuda:
                native.set_forward_waits([])
            if hasattr(native, "enable_wire_grads"):
                native.enable_wire_grads(False)

    def finish(self, opt_stream=None):
        """the compute stream waits for every outstanding collective and for the optimizer stream (no host sync on GPU)"""
        opt_stream = opt_stream if opt_stream is not None else self._last_opt
        if self._deferred or self._repl_keys:    # a step whose gathers were never issued (gather_deferred not called): do it now
            self.gather_deferred(self._native, opt_stream)
        for w in self.pending:
            w.wait()
        self.pending = []
        if opt_stream is not None:
            torch.cuda.current_stream().wait_stream(opt_stream)

    def wait(self):
        self.finish()

    # ---- parameters ----------------------------------------------------------------------------------------------------------
    def broadcast_parameters(self, native, src=0):
        """DDP's start-up broadcast rank0 -> all"""
        if not self.active:
            return
        dist.broadcast(native.W, src=src, group=self.group)
        dist.broadcast(native.P, src=src, group=self.group)
        if getattr(native, "Wlp", None) is not None:
            native.sync_lowp(force=True)

    def sync_tune_table(self, src=0, force=False):
        """Every rank adopts rank `src`'s table of tuned GEMM configurations (a collective).  Ranks that tune on their own can
        settle on different tiles for the same product — a rank with a slower pick is then the straggler of every later step
         — so in a data-parallel job ONLY rank `src` tunes in situ (`lead_tuning()` switches the others'
        tuner off) and the others replace their table with its table here.  TrainLoop calls this after optimizer steps 1, 2, 4,
        ..., 256 and then every 256: unseen signatures (a new bucket of `t`) keep appearing for a while; between two calls a
        follower launches its heuristic configuration for them (same results, possibly another speed).
        The text only travels when some rank's table differs from the one last adopted: one int64 flag per rank,
        MAX-reduced on the device - rank `src` raises it when its table changed since the last call, a follower when ITS table is
        no longer the adopted one (a local `tune_table_merge` / cache import) - then, only if it is up, the pickled text from
        `src`.  With the shipped table the normal case is "unchanged": one 8-byte all-reduce and one host read per call.
        Returns the table text (None when nothing was sent)."""
        from . import _lib
        if not self.active:
            return _lib.tune_table_text()
        text = _lib.tune_table_text()
        flag = 1 if (force or text != getattr(self, "_tune_sent", None)) else 0
        nccl = dist.get_backend(self.group) == "nccl"
        head = torch.tensor([flag], dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()) if nccl else "cpu")
        dist.all_reduce(head, op=dist.ReduceOp.MAX, group=self.group)
        if int(head.item()) == 0:
            return None
        box = [text if self.rank == src else None]
        dist.broadcast_object_list(box, src=src, group=self.group)
        if self.rank != src:
            _lib.tune_table_merge(box[0], replace=True)
        self._tune_sent = box[0]
        return box[0]

    def lead_tuning(self, src=0):
        """only rank `src` times GEMM candidates in situ; the other ranks take its choices (sync_tune_table)"""
        from . import _lib
        if self.active and self.rank != src:
            _lib.load().mebt_gemm_autotune(0)

    @staticmethod
    def tune_sync_due(step):
        """optimizer steps after which the followers adopt the lead's table: 1, 2, 4, ..., 256, then every 256 (each sync point is a
        host read that drains every rank's queue: ADVICE r05 - unseen signatures appear early, so the early points are dense)"""
        return step > 0 and (((step & (step - 1)) == 0 and step <= 256) or step % 256 == 0)

    def consolidate(self, native, optimizer_state=False):
        """Make the fp32 masters (and, on request, the AdamW moments) complete on every rank: after sharded steps each rank
        holds the current values of its own shards only.  A collective: call on ALL ranks (before state_dict / a checkpoint)."""
        if not self.active or self.mode != "sharded":
            return
        self.finish()                            # deferred gathers of the last step, the optimizer stream
        world, rank = self.world_size, self.rank
        tensors = [(0, native.W)] if self.master_stale else []
        if optimizer_state and native.adam is not None:
            mW, vW, mP, vP = native.adam
            tensors += [(0, mW), (0, vW), (1, mP), (1, vP)]
        for which_t, t in tensors:
            for which, a, b in sorted(self._sharded_ranges):        # the same cuts the optimizer used (replicated buckets are complete)
                if which != which_t:
                    continue
                shard = (b - a) // world
                self._all_gather(t[a:b], t[a + rank * shard:a + (rank + 1) * shard]).wait()
        self.master_stale = False

    def mean_scalars(self, t):
        """one small all-reduce for the logged scalars (the reference issues four: loss, acc1, acc5, lr)"""
        if not self.active:
            return t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world_size
