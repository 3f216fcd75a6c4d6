#!/usr/bin/env python3
"""The grouped weight-gradient launch (mebt_op_wgrad_grouped) at the Sky-16f shapes (B = 6: 3072 context / target rows, 1536 latent
rows, 4608 = latents + targets): stored gradients vs AdamW in the epilogue, per block as the step launches it and for PAIRS of
consecutive blocks in one launch (1536 tiles of 128 x 128 fill the chip in whole rounds; 768 do not), over forced tile shapes / ring
depths (mebt_debug_grouped_config).  (Round 6 also claimed the tiles from per-XCD two-ended queues so that CUs / XCDs of odd parity
start at the short reductions - slower in every form, removed: profiles/r06_wgrad_overlap_evidence.txt.)
GPU box: python tools/wgrad_bench.py [--cases enc,dec+lt2l,...]"""
import argparse
import ctypes as C
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--cases", default="enc,dec,self,lt2l,dec+lt2l,self+enc,dec+enc")
ap.add_argument("--configs", default="0,128x64x2,128x128x2,128x128x3,64x128x2,128x64x3")
args = ap.parse_args()
lib = _lib.load()
d = 1024
C3, L, T46 = 3072, 1536, 4608


def block(kind):
    """the items of engine.cpp:backward_layer (n_out, k_in, tokens): fc2, fc1, proj, then q and k|v (or the fused q|k|v)"""
    rows_q = {"enc": L, "self": L, "lt2l": L, "dec": C3}[kind]
    rows_k = {"enc": C3, "self": L, "lt2l": T46, "dec": L}[kind]
    it = [("fc2", d, 4 * d, rows_q), ("fc1", 4 * d, d, rows_q), ("proj", d, d, rows_q)]
    if kind == "self":
        it.append(("qkv", 3 * d, d, rows_q))
    else:
        it += [("q", d, d, rows_q), ("kv", 2 * d, d, rows_k)]
    return it


pool = 6            # rotate parameter sets so that p / m / v are HBM-cold like in the step
modes = [(0, "")]
configs = [tuple(int(v) for v in c.split("x")) if c != "0" else None for c in args.configs.split(",")]
for case in args.cases.split(","):
    items = [it for k in case.split("+") for it in block(k)]
    n = len(items)
    tot = sum(a * b for _, a, b, _ in items)
    offs, o = [], 0
    for _, a, b, _ in items:
        offs.append(o)
        o += a * b
    dYs = [torch.randn(T, a, device="cuda").to(torch.bfloat16) for _, a, b, T in items]
    Xs = [torch.randn(T, b, device="cuda").to(torch.bfloat16) for _, a, b, T in items]
    Ws = [torch.randn(tot, device="cuda") * 0.02 for _ in range(pool)]
    ms = [torch.zeros(tot, device="cuda") for _ in range(pool)]
    vs = [torch.zeros(tot, device="cuda") for _ in range(pool)]
    lps = [w.to(torch.bfloat16) for w in Ws]
    gW = torch.zeros(tot, device="cuda")
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    i32 = lambda v: (C.c_int32 * n)(*v)
    fl = sum(2.0 * T * a * b for _, a, b, T in items)
    nb = len(case.split("+"))
    print(f"# {case}: {n} items, {tot / 1e6:.2f} M parameters, {fl / 1e9:.1f} GFLOP, {tot * 26 / 1e6:.0f} MB of optimizer traffic  (us below are PER BLOCK: launch / {nb})")
    for cfgc in configs:
        ref_gW = None
        if cfgc is None:
            lib.mebt_debug_grouped_config(0, 0, 0)
            cname = "table / heuristic"
        else:
            lib.mebt_debug_grouped_config(*cfgc)
            cname = "%d x %d ring %d" % cfgc
        for mode, mname in modes:
            res = []
            for fused in (0, 1):
                ctr = [0]

                def run():
                    i = ctr[0] % pool
                    ctr[0] += 1
                    check(lib.mebt_op_wgrad_grouped(n, arr(dYs), arr(Xs), i32([a for _, a, b, _ in items]), i32([b for _, a, b, _ in items]), i32([T for _, _, _, T in items]),
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
                if not fused:                           # whatever tile shape / claim order: the same reduction order per element, bit for bit
                    if ref_gW is None:
                        ref_gW = gW.clone()
                    assert torch.equal(gW, ref_gW), (case, cname, mname)
            print(f"  {cname + mname:40s} stored {res[0] / nb:7.1f} us ({fl / res[0] / 1e6:5.0f} TF/s)   +AdamW {res[1] / nb:7.1f} us  (delta {(res[1] - res[0]) / nb:6.1f} us = "
                  f"{tot * 26 / max(res[1] - res[0], 1e-9) / 1e6:5.2f} TB/s of optimizer traffic in the delta)", flush=True)
    lib.mebt_debug_grouped_config(0, 0, 0)
    del dYs, Xs, Ws, ms, vs, lps, gW
    torch.cuda.empty_cache()
