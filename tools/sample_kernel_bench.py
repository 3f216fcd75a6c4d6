#!/usr/bin/env python3
"""Time of the draw kernel alone (mebt_op_sample_seeded / mebt_op_sample_scatter) on [rows, 16384] logits: top-k on / off, probability map
written or not.  Bytes: rows x 64 KiB read (+ the same written with the map).  MEBT_SAMPLE_FAST=0 times the LDS-row kernel it replaced."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream
lib = _lib.load()
V = 16384
for rows in (1024, 32768):
    lg = torch.randn(rows, V, device="cuda") * 2.0
    ids = torch.empty(rows, dtype=torch.long, device="cuda")
    sc = torch.empty(rows, device="cuda")
    probs = torch.empty(rows, V, device="cuda")
    lb = lg.bfloat16()
    for k, wp, bf in ((0, False, 0), (32, False, 0), (0, True, 0), (32, True, 0), (0, False, 1), (32, False, 1), (0, False, 2), (32, False, 2), (0, False, 3), (32, False, 3)):
        if bf:      # 1: the head's bf16 logits (mebt_op_sample_lp), arg-max p / q; 2: fp32 logits, inverse-CDF draw; 3: bf16 logits + inverse CDF (the loops' production draw)
            fn = lambda: check(lib.mebt_op_sample_lp(ptr(lb) if bf != 2 else ptr(lg), 0 if bf == 2 else 1, None, 1234, 1.0, k, ptr(ids), ptr(sc), None, None, 1, rows, rows, V,
                                                     1 if bf >= 2 else 0, cur_stream()))
        else:
            fn = lambda: check(lib.mebt_op_sample_seeded(ptr(lg), 1234, 1.0, k, 0.0, ptr(ids), ptr(sc), ptr(probs) if wp else None, rows, V, cur_stream()))
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        gb = rows * V * (2 if bf in (1, 3) else 4) * (2 if wp else 1) / 1e9
        print(f"rows {rows:6d} top_k {k:3d} probs {int(wp)} {'bf16' if bf in (1, 3) else 'fp32'} logits{' icdf' if bf >= 2 else '     '}: {us:9.1f} us  {gb / us * 1e6:7.1f} GB/s moved  ({rows * V * 4 / us * 1e-3:7.1f} GB/s fp32-equivalent)")
