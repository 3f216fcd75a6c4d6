#!/usr/bin/env python3
"""GEMM time as a function of K (fixed cost vs main loop), with and without the C store.
Run under `rocprofv3 --kernel-trace` and read the durations with tools/kscan_report.py."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream
lib = _lib.load()
def t(M, N, K, akc=1, bkc=1, iters=30, c32=0):
    A = torch.randn(M, K, device='cuda').bfloat16() if akc else torch.randn(K, M, device='cuda').bfloat16()
    B = torch.randn(N, K, device='cuda').bfloat16() if bkc else torch.randn(K, N, device='cuda').bfloat16()
    C = torch.empty(M, N, device='cuda', dtype=torch.float32 if c32 else torch.bfloat16)
    lda, ldb = A.shape[1], B.shape[1]
    def run(): check(lib.mebt_op_gemm(_lib.BF16, ptr(A), ptr(B), ptr(C), None, None, None, M, N, K, lda, ldb, N, N, akc, bkc, 0, c32, 0, 1, cur_stream()))
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for nostore in (0, 1):
    lib.mebt_debug_gemm_variant(199 if nostore else -1)
    for (M, N) in ((1536, 1024), (3072, 1024), (1536, 4096), (3072, 4096)):
        r = []
        for K in (64, 256, 1024, 4096):
            r.append(f"K={K}:{t(M, N, K):6.1f}")
        print("nostore" if nostore else "store  ", M, N, "  ".join(r))
