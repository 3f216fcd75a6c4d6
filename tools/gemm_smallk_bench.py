#!/usr/bin/env python3
"""The small products of a step (1536 x 1024 x 1024 and relatives: one chip's worth of small tiles, 16 k-tiles): plain ring, pipelined loop and two
pipelines per workgroup, cold operands, next to torch.matmul.  GPU box: python tools/gemm_smallk_bench.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream

lib = _lib.load()
dt, tt = _lib.BF16, torch.bfloat16
scratch = torch.empty((384 + 128) << 20, dtype=torch.uint8, device="cuda")
lib.mebt_debug_gemm_scratch(ptr(scratch), scratch.numel())
d = 1024
shapes = [("fwd q/proj M1536", 1536, d, d, 1), ("dgrad q/proj M1536", 1536, d, d, 0), ("fwd q/proj M3072", 3072, d, d, 1), ("dgrad q/proj M3072", 3072, d, d, 0),
          ("fwd kv dec M1536", 1536, 2 * d, d, 1), ("dgrad kv M1536 K2048", 1536, d, 2 * d, 0)]
variants = [((96, 64), 2), ((96, 64), 4), ((96, 64), 12), ((96, 64), 18), ((96, 64), 19), ((64, 64), 3), ((64, 64), 19),
            ((96, 128), 3), ((128, 64), 19)]
for label, M, N, K, bkc in shapes:
    pool = max(1, int(6e8 // ((M * K + N * K) * 2)))
    A = torch.randn(pool, M, K, device="cuda").to(tt)
    B = torch.randn((pool,) + ((N, K) if bkc else (K, N)), device="cuda").to(tt)
    C = torch.empty(M, N, device="cuda", dtype=tt)
    ldb = B.shape[2]
    ctr = [0]

    def run():
        i = ctr[0] % pool
        ctr[0] += 1
        check(lib.mebt_op_gemm(dt, A[i].data_ptr(), B[i].data_ptr(), ptr(C), None, None, None, M, N, K, K, ldb, N, N, 1, bkc, 0, 0, 0, 1, cur_stream()))

    def run_torch():
        i = ctr[0] % pool
        ctr[0] += 1
        torch.matmul(A[i], B[i].t() if bkc else B[i], out=C)

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

    res = []
    for tile, code in variants:
        lib.mebt_debug_gemm_tile(*tile)
        lib.mebt_debug_gemm_variant(code)
        res.append((timeit(run), tile, code))
    lib.mebt_debug_gemm_tile(0, 0)
    lib.mebt_debug_gemm_variant(-1)
    tuned = timeit(run)
    tq = timeit(run_torch)
    print(f"{label}: tuned {tuned:.1f} us, torch {tq:.1f} us; forced: " + "  ".join(f"{t[0]}x{t[1]}/{c}: {us:.1f}" for us, t, c in res), flush=True)
