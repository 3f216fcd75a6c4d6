#!/usr/bin/env python3
"""Large square GEMMs (where tile quantisation and fixed costs vanish): TF/s of each block tile, forced."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream
lib = _lib.load()
def t(M, N, K, bkc, iters=10):
    A = torch.randn(M, K, device='cuda').bfloat16()
    B = (torch.randn(N, K, device='cuda') if bkc else torch.randn(K, N, device='cuda')).bfloat16()
    C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    def run(): check(lib.mebt_op_gemm(_lib.BF16, ptr(A), ptr(B), ptr(C), None, None, None, M, N, K, K, B.shape[1], N, N, 1, bkc, 0, 0, 0, 1, cur_stream()))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 4096), (3072, 16384, 1024), (3072, 4096, 1024)):
    for bkc in (1, 0):
        r = []
        for tile, variants in (((256, 256), (2,)), ((192, 128), (2, 3)), ((128, 128), (2, 3, 4))):
            lib.mebt_debug_gemm_tile(*tile)
            for v in variants:
                lib.mebt_debug_gemm_variant(v)
                us = t(M, N, K, bkc)
                r.append(f"{tile[0]}x{tile[1]}/r{v}: {2.0 * M * N * K / us / 1e6:5.0f}")
        lib.mebt_debug_gemm_tile(0, 0); lib.mebt_debug_gemm_variant(-1)
        print(f"M={M} N={N} K={K} B {'KC' if bkc else 'RC'}:  " + "  ".join(r))
