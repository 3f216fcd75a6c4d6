#!/usr/bin/env python3
"""Plain LDS-DMA main loop (gemm_tile_dma) vs the software-pipelined one (gemm_tile_pipe) per tile / ring depth on the
GEMM shapes of one Sky-16f train step.  GPU box: python tools/gemm_pipe_bench.py [--cold]"""
import argparse
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--cold", action="store_true")
ap.add_argument("--codes", default="2,3,4,10,11,12")
args = ap.parse_args()
lib = _lib.load()
dt, tt = _lib.BF16, torch.bfloat16
d = 1024
shapes = [("fwd proj M1536", 1536, d, d, 1, 1), ("fwd fc1 M1536", 1536, 4 * d, d, 1, 1), ("fwd fc2 M1536", 1536, d, 4 * d, 1, 1),
          ("dgrad proj M1536", 1536, d, d, 1, 0), ("dgrad fc2 M1536", 1536, 4 * d, d, 1, 0), ("dgrad fc1 M1536", 1536, d, 4 * d, 1, 0),
          ("fwd fc1 M3072", 3072, 4 * d, d, 1, 1), ("fwd fc2 M3072", 3072, d, 4 * d, 1, 1), ("fwd kv enc", 3072, 2 * d, d, 1, 1),
          ("wgrad fc1 M1536", 4 * d, d, 1536, 0, 0), ("wgrad proj M1536", d, d, 1536, 0, 0)]
codes = [int(c) for c in args.codes.split(",")]
tiles = ((128, 128), (192, 128), (96, 128), (96, 64), (128, 64), (64, 128), (64, 64))
print("columns per tile: " + " / ".join(f"{'pipe ' if c >= 8 else 'ring '}{c - 8 if c >= 8 else c}" for c in codes) + "  (us)")
for label, M, N, K, akc, bkc in shapes:
    pool = max(1, int(6e8 // ((M * K + N * K) * 2))) if args.cold else 1
    A = torch.randn((pool,) + ((M, K) if akc else (K, M)), device="cuda").to(tt)
    B = torch.randn((pool,) + ((N, K) if bkc else (K, N)), device="cuda").to(tt)
    cf32 = 0 if akc else 1
    C = torch.empty(M, N, device="cuda", dtype=torch.float32 if cf32 else tt)
    lda, ldb = A.shape[2], B.shape[2]
    ctr = [0]

    def run():
        i = ctr[0] % pool
        ctr[0] += 1
        check(lib.mebt_op_gemm(dt, A[i].data_ptr(), B[i].data_ptr(), ptr(C), None, None, None, M, N, K, lda, ldb, N, N, akc, bkc, 0, cf32, 0, 1, cur_stream()))

    def timeit():
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.iters

    out = []
    best = (1e9, None)
    for bm, bn in tiles:
        if M % bm and not akc:
            pass
        lib.mebt_debug_gemm_tile(bm, bn)
        r = []
        for c in codes:
            ring = c - 8 if c >= 8 else c
            if ring * (bm + bn) * 128 > (160 if c >= 8 else 128) * 1024:
                r.append("   -")
                continue
            lib.mebt_debug_gemm_variant(c)
            us = timeit()
            r.append(f"{us:5.1f}")
            if us < best[0]:
                best = (us, f"{bm}x{bn} code {c}")
        out.append(f"{bm}x{bn}: " + "/".join(r))
    lib.mebt_debug_gemm_tile(0, 0)
    lib.mebt_debug_gemm_variant(-1)
    fl = 2.0 * M * N * K
    print(f"{label:18s} {M}x{N}x{K}  best {best[0]:6.1f} us ({fl / best[0] / 1e6:5.0f} TF/s) {best[1]}\n     " + "\n     ".join(out))
