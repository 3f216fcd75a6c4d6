#!/usr/bin/env python3
"""The grouped weight-gradient launch of one block (mebt_op_wgrad_grouped) at the Sky-16f shapes, stored gradients vs AdamW in the
epilogue, whole block vs subsets of its items (how does the launch time scale with the number of tile rounds?).
GPU box: python tools/wgrad_bench.py [--tokens 1536]"""
import argparse
import ctypes as C
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=1536)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
lib = _lib.load()
d, T = 1024, args.tokens
block = [("q", d, d), ("k", d, d), ("v", d, d), ("proj", d, d), ("fc1", 4 * d, d), ("fc2", d, 4 * d)]
cases = {"whole block (768 tiles of 128x128)": block, "fc1 + fc2 (512 tiles)": block[4:], "q k v proj (256 tiles)": block[:4],
         "fc1 (256 tiles)": block[4:5], "q k v proj fc1 (512 tiles)": block[:5], "fc1 fc2 q k (640 tiles)": block[4:] + block[:2]}
pool = 6            # rotate parameter sets so that p / m / v are HBM-cold like in the step (6 x 327 MB)
for name, items in cases.items():
    n = len(items)
    tot = sum(a * b for _, a, b in items)
    offs, o = [], 0
    for _, a, b in items:
        offs.append(o)
        o += a * b
    dYs = [torch.randn(T, a, device="cuda").to(torch.bfloat16) for _, a, b in items]
    Xs = [torch.randn(T, b, device="cuda").to(torch.bfloat16) for _, a, b in items]
    Ws = [torch.randn(tot, device="cuda") * 0.02 for _ in range(pool)]
    ms = [torch.zeros(tot, device="cuda") for _ in range(pool)]
    vs = [torch.zeros(tot, device="cuda") for _ in range(pool)]
    lps = [w.to(torch.bfloat16) for w in Ws]
    gW = torch.zeros(tot, device="cuda")
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    i32 = lambda v: (C.c_int32 * n)(*v)
    res = []
    for fused in (0, 1):
        ctr = [0]

        def run():
            i = ctr[0] % pool
            ctr[0] += 1
            check(lib.mebt_op_wgrad_grouped(n, arr(dYs), arr(Xs), i32([a for _, a, b in items]), i32([b for _, a, b in items]), i32([T] * n),
                                            (C.c_int64 * n)(*offs), None, ptr(Ws[i]), ptr(gW), ptr(ms[i]), ptr(vs[i]), ptr(lps[i]), fused,
                                            1e-5, 0.9, 0.95, 1e-8, 0.01, 1, 1.0, cur_stream()))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / args.iters)
    fl = 2.0 * T * tot
    print(f"{name:40s} {tot / 1e6:6.2f} M params  stored {res[0]:7.1f} us ({fl / res[0] / 1e6:5.0f} TF/s)   +AdamW {res[1]:7.1f} us  (delta {res[1] - res[0]:6.1f} us, "
          f"{tot * 26 / (res[1] - res[0]) / 1e6 if res[1] > res[0] else 0:5.2f} TB/s of optimizer traffic in the delta)")
    del dYs, Xs, Ws, ms, vs, lps, gW
    torch.cuda.empty_cache()
