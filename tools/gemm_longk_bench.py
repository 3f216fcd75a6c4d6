#!/usr/bin/env python3
"""The K = 4096 products with one chip's worth of tiles (MLP down-projection and its dgrad): every forced (tile, staging)
variant incl. the split-K ones, cold operands, next to torch.matmul.  GPU box: python tools/gemm_longk_bench.py"""
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
shapes = [("fwd fc2 M3072", 3072, d, 4 * d, 1), ("dgrad fc1 M3072", 3072, d, 4 * d, 0), ("fwd fc2 M1536", 1536, d, 4 * d, 1), ("dgrad fc1 M1536", 1536, d, 4 * d, 0),
          ("dgrad kv enc", 3072, d, 2 * d, 0)]
variants = [((96, 128), 3), ((96, 128), 11), ((96, 128), 18), ((96, 128), 34), ((96, 128), 35), ((96, 128), 67), ((128, 128), 3), ((128, 128), 35), ((128, 128), 67),
            ((192, 128), 35), ((96, 64), 3), ((96, 64), 19), ((96, 64), 35), ((128, 64), 19), ((64, 128), 19)]
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

    def timeit(fn, iters=20):
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
    res.sort()
    print(f"{label}: tuned {tuned:.1f} us, torch {tq:.1f} us; forced: " + "  ".join(f"{t[0]}x{t[1]}/{c}: {us:.1f}" for us, t, c in res[:8]), flush=True)
