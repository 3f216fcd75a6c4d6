#!/usr/bin/env python3
"""Per-shape timing of the GEMM kernel on the shapes of one Sky-16f train step (B=6, NC=NT=512).
Usage (GPU box): python tools/gemm_bench.py [--dtype bf16]"""
import argparse
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--variants", action="store_true")
ap.add_argument("--torch", action="store_true", help="calibration column: torch.matmul (hipBLASLt/rocBLAS) on the same operands")
ap.add_argument("--cold", action="store_true", help="rotate operands through a >600 MB pool (HBM-cold, like in the real step)")
args = ap.parse_args()
lib = _lib.load()
dt = _lib.BF16 if args.dtype == "bf16" else _lib.F32
tt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
d = 1024
# (label, count per step, M, N, K, a_kc, b_kc, c_f32, split)
shapes = []
for tag, M, cnt in (("M1536", 1536, 18), ("M3072", 3072, 6)):
    shapes += [(f"fwd q/proj {tag}", cnt * 2, M, d, d, 1, 1, 0, 1), (f"fwd fc1 {tag}", cnt, M, 4 * d, d, 1, 1, 0, 1),
               (f"fwd fc2 {tag}", cnt, M, d, 4 * d, 1, 1, 0, 1),
               (f"dgrad q/proj {tag}", cnt * 2, M, d, d, 1, 0, 0, 1), (f"dgrad fc2->d4 {tag}", cnt, M, 4 * d, d, 1, 0, 0, 1),
               (f"dgrad fc1->dh {tag}", cnt, M, d, 4 * d, 1, 0, 0, 1),
               (f"wgrad q/proj {tag}", cnt * 2, d, d, M, 0, 0, 1, 0), (f"wgrad fc1 {tag}", cnt, 4 * d, d, M, 0, 0, 1, 0),
               (f"wgrad fc2 {tag}", cnt, d, 4 * d, M, 0, 0, 1, 0)]
shapes += [("fwd qkv self", 6, 1536, 3 * d, d, 1, 1, 0, 1), ("fwd kv enc", 7, 3072, 2 * d, d, 1, 1, 0, 1),
           ("fwd kv lt2l", 5, 4608, 2 * d, d, 1, 1, 0, 1), ("fwd kv dec", 6, 1536, 2 * d, d, 1, 1, 0, 1),
           ("dgrad kv enc", 7, 3072, d, 2 * d, 1, 0, 0, 1), ("wgrad kv enc", 7, 2 * d, d, 3072, 0, 0, 1, 0),
           ("fwd head", 1, 3072, 16384, d, 1, 1, 1, 1), ("dgrad head", 1, 3072, d, 16384, 1, 0, 0, 1),
           ("wgrad head", 1, 16384, d, 3072, 0, 0, 1, 0)]
tot_ms = tot_fl = 0.0
print(f"{'shape':24s} {'cnt':>3s} {'M':>6s} {'N':>6s} {'K':>6s}  {'us':>8s} {'TF/s':>7s}")
for label, cnt, M, N, K, akc, bkc, cf32, split in shapes:
    esz = 2 if args.dtype == "bf16" else 4
    pool = max(1, int(6e8 // ((M * K + N * K) * esz))) if args.cold else 1
    A = torch.randn((pool,) + ((M, K) if akc else (K, M)), device="cuda").to(tt)
    B = torch.randn((pool,) + ((N, K) if bkc else (K, N)), device="cuda").to(tt)
    C = torch.empty(M, N, device="cuda", dtype=torch.float32 if (cf32 or dt == _lib.F32) else tt)
    lda, ldb = A.shape[2], B.shape[2]
    ctr = [0]
    def run():
        i = ctr[0] % pool
        ctr[0] += 1
        check(lib.mebt_op_gemm(dt, A[i].data_ptr(), B[i].data_ptr(), ptr(C), None, None, None, M, N, K, lda, ldb, N, N, akc, bkc, 0, cf32, 0, split, cur_stream()))
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
    lib.mebt_debug_gemm_variant(-1)
    us = timeit()
    fl = 2.0 * M * N * K
    tot_ms += cnt * us * 1e-3
    tot_fl += cnt * fl
    var = ""
    if args.dtype == "bf16" and args.variants:
        for bm, bn in ((128, 128), (192, 128), (96, 128), (96, 64), (128, 64), (64, 128), (64, 64)):
            lib.mebt_debug_gemm_tile(bm, bn)
            r = []
            for v in ((2, 3, 4) if bm in (96, 192) else (0, 2, 3, 4, 5)):    # register-staged / LDS-DMA with 2,3,4,5 ring stages
                lib.mebt_debug_gemm_variant(v)
                r.append(f"{fl / timeit() / 1e6:4.0f}")
            var += f" |{bm}x{bn} " + "/".join(r)
        lib.mebt_debug_gemm_tile(0, 0)
        lib.mebt_debug_gemm_variant(-1)
    if args.torch:
        Ct = torch.empty(M, N, device="cuda", dtype=tt)
        def run():
            i = ctr[0] % pool
            ctr[0] += 1
            a = A[i] if akc else A[i].t()
            b = B[i].t() if bkc else B[i]
            torch.matmul(a, b, out=Ct)
        ust = timeit()
        tot_t = globals().get("tot_t", 0.0) + cnt * ust * 1e-3
        globals()["tot_t"] = tot_t
        var += f" | torch {ust:7.1f} us {fl / ust / 1e6:6.0f} TF/s"
    print(f"{label:24s} {cnt:3d} {M:6d} {N:6d} {K:6d}  {us:8.1f} {fl / us / 1e6:7.1f}{var}")
if args.torch:
    print(f"torch.matmul step total: {globals()['tot_t']:.2f} ms")
print(f"step total: {tot_ms:.2f} ms, {tot_fl / 1e12:.2f} TFLOP -> {tot_fl / tot_ms / 1e9:.0f} TF/s")
