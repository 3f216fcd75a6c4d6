#!/usr/bin/env python3
"""Experiment: does running the two halves of the batch as two independent chains on two streams (each kernel half the
size, always another kernel ready to fill launch / prologue / epilogue bubbles) beat one chain over the whole batch?
Eval forward of the Sky-16f network: B = 6 on one stream vs 2 x (B = 3) on two streams sharing the same weights."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import presets
from mebt_amd.engine import NativeModel

cfg = presets.sky_16f(dropout=0.0)
torch.manual_seed(0)
model = presets.build_model(cfg, compute_dtype="bf16").cuda().eval()
nm = model._ensure_native()
p = cfg.model.params
modes = [b.mode for b in model.transformer.blocks]
twins = []
for _ in range(2):
    t = NativeModel(p.n_layer, p.n_head, p.n_embd, 16384, p.sos_emb, p.block_size, modes, dtype="bf16")
    t.device = nm.device
    t.W, t.P, t.Wlp = nm.W, nm.P, nm.Wlp
    t.bind()
    t._w_version = (t.W._version, 0)
    twins.append(t)
g = torch.Generator().manual_seed(1)
x = torch.randint(0, 16384, (6, 1024), generator=g).cuda()
idx = torch.stack([torch.randperm(1024, generator=g) for _ in range(6)]).cuda()
ci, ti = idx[:, :512].contiguous(), idx[:, 512:].contiguous()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
halves = [(x[:3].contiguous(), ci[:3].contiguous(), ti[:3].contiguous()), (x[3:].contiguous(), ci[3:].contiguous(), ti[3:].contiguous())]


def whole():
    nm.forward(x, ci, ti, training=False)


def split():
    cur = torch.cuda.current_stream()
    for s, t, h in zip(streams, twins, halves):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            t.forward(*h, training=False)
    for s in streams:
        cur.wait_stream(s)


def serial_halves():
    for t, h in zip(twins, halves):
        t.forward(*h, training=False)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, fn in (("B=6 one stream", whole), ("2 x B=3 two streams", split), ("2 x B=3 one stream", serial_halves), ("B=6 one stream", whole)):
    print(f"{name:24s} {timed(fn):.3f} ms")
