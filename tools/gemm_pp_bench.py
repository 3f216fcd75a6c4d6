#!/usr/bin/env python3
"""The 256 x 256 tile as one lock-step 8-wave group (variant 2) vs two staggered 4-wave groups (variant 9,
gemm_bf16_pp_kernel) vs the best 4-wave tile vs torch.matmul (hipBLASLt) on the large products of configs 2 and 4.
GPU box: python tools/gemm_pp_bench.py [--cold]"""
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
args = ap.parse_args()
lib = _lib.load()
dt, tt = _lib.BF16, torch.bfloat16
d = 1024
shapes = [("fwd fc1 M3072", 3072, 4 * d, d, 1), ("fwd fc2 M3072", 3072, d, 4 * d, 1), ("fwd kv enc M3072", 3072, 2 * d, d, 1),
          ("head fwd M3072", 3072, 16384, d, 1), ("head dgrad M3072", 3072, d, 16384, 0), ("dgrad fc2 M3072", 3072, 4 * d, d, 0),
          ("fwd fc1 M1536", 1536, 4 * d, d, 1), ("c4 kv M31744", 31744, 2 * d, d, 1), ("c4 fc1 M4096", 4096, 4 * d, d, 1),
          ("c4 head M31744", 31744, 16384, d, 1)]
for label, M, N, K, bkc in shapes:
    pool = max(1, int(6e8 // ((M * K + N * K) * 2))) if args.cold else 1
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

    def timeit(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.iters

    res = {}
    for name, tile, code in (("w8", (256, 256), 2), ("pp", (256, 256), 9), ("192x128 r3", (192, 128), 3), ("128x128 r3", (128, 128), 3), ("128x128 pipe3", (128, 128), 11)):
        lib.mebt_debug_gemm_tile(*tile)
        lib.mebt_debug_gemm_variant(code)
        res[name] = timeit(run)
        if name == "pp":
            ref = torch.matmul(A[(ctr[0] - 1) % pool].float(), (B[(ctr[0] - 1) % pool].float().t() if bkc else B[(ctr[0] - 1) % pool].float()))
            err = (C.float() - ref).abs().max().item() / ref.abs().max().item()
            res["pp_err"] = err
    lib.mebt_debug_gemm_tile(0, 0)
    lib.mebt_debug_gemm_variant(-1)
    res["torch"] = timeit(run_torch)
    tf = 2.0 * M * N * K / 1e6
    print(f"{label:20s} " + "  ".join(f"{k} {v:7.1f} us ({tf / v:5.0f} TF/s)" if k != "pp_err" else f"[pp rel err {v:.1e}]" for k, v in res.items()), flush=True)
