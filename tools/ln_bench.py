#!/usr/bin/env python3
"""LayerNorm forward / backward at the row counts of a Sky-16f train step (d = 1024, bf16), cold operands (a pool larger than
the caches), next to a plain device copy of the same number of bytes.  GPU box: python tools/ln_bench.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream

lib = _lib.load()
d, tt = 1024, torch.bfloat16
g = torch.ones(d, device="cuda")
b = torch.zeros(d, device="cuda")


def timeit(fn, iters=40):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for rows in (1536, 3072, 4608):
    pool = max(2, int(6e8 // (rows * d * 2 * 3)))
    x = torch.randn(pool, rows, d, device="cuda").to(tt)
    dy = torch.randn(pool, rows, d, device="cuda").to(tt)
    y = torch.empty(pool, rows, d, device="cuda", dtype=tt)
    mean = torch.empty(rows, device="cuda")
    rstd = torch.empty(rows, device="cuda")
    dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    c = [0]

    def fwd():
        i = c[0] % pool
        c[0] += 1
        check(lib.mebt_op_layernorm_fwd(_lib.BF16, x[i].data_ptr(), y[i].data_ptr(), ptr(g), ptr(b), ptr(mean), ptr(rstd), rows, d, cur_stream()))

    def bwd():
        i = c[0] % pool
        c[0] += 1
        check(lib.mebt_op_layernorm_bwd(_lib.BF16, x[i].data_ptr(), dy[i].data_ptr(), ptr(g), ptr(mean), ptr(rstd), y[i].data_ptr(), ptr(dg), ptr(db), rows, d, cur_stream()))

    def copy1():
        i = c[0] % pool
        c[0] += 1
        y[i].copy_(x[i])

    def copy2():   # two reads, one write: the backward's minimum
        i = c[0] % pool
        c[0] += 1
        torch.add(x[i], dy[i], out=y[i])

    mb = rows * d * 2 / 1e6
    tf, tb, t1, t2 = timeit(fwd), timeit(bwd), timeit(copy1), timeit(copy2)
    print(f"rows {rows}: fwd {tf:6.1f} us ({2 * mb / tf:5.2f} TB/s)   bwd {tb:6.1f} us ({3 * mb / tb:5.2f} TB/s of x, dy, dx)   "
          f"copy {t1:6.1f} us ({2 * mb / t1:5.2f} TB/s)   x + dy -> y {t2:6.1f} us ({3 * mb / t2:5.2f} TB/s)", flush=True)
