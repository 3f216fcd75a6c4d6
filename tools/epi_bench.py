#!/usr/bin/env python3
"""Cost of the fused epilogues on the two GEMM shapes that carry them (cold operands)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream
lib = _lib.load()
d = 1024
def bench(label, M, N, K, akc, bkc, epi, with_bias):
    pool = max(1, int(6e8 // ((M * K + N * K) * 2)))
    A = torch.randn((pool,) + ((M, K) if akc else (K, M)), device="cuda").bfloat16()
    B = torch.randn((pool,) + ((N, K) if bkc else (K, N)), device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); C2 = torch.empty_like(C)
    aux = torch.randn(M, N, device="cuda").bfloat16(); bias = torch.randn(N, device="cuda")
    lda, ldb = A.shape[2], B.shape[2]
    ctr = [0]
    def run():
        i = ctr[0] % pool; ctr[0] += 1
        check(lib.mebt_op_gemm(1, A[i].data_ptr(), B[i].data_ptr(), ptr(C), ptr(C2) if epi == 1 else None, ptr(bias) if with_bias else None,
                               ptr(aux) if epi in (2, 3) else None, M, N, K, lda, ldb, N, N, akc, bkc, epi, 0, 0, 1, cur_stream()))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    print(f"{label:40s} {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF/s")
for M in (1536, 3072):
    bench(f"fwd fc1 M{M} plain", M, 4 * d, d, 1, 1, 0, False)
    bench(f"fwd fc1 M{M} bias+GELU (pre+act out)", M, 4 * d, d, 1, 1, 1, True)
    bench(f"dgrad fc2->d4 M{M} plain", M, 4 * d, d, 1, 0, 0, False)
    bench(f"dgrad fc2->d4 M{M} * gelu'(pre)", M, 4 * d, d, 1, 0, 3, False)
    bench(f"fwd proj M{M} plain", M, d, d, 1, 1, 0, False)
    bench(f"fwd proj M{M} bias+resid", M, d, d, 1, 1, 2, True)
