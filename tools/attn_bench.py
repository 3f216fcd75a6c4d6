#!/usr/bin/env python3
"""Isolated timing of the MFMA attention kernels on the four Sky-16f routings (B=6, H=16, hd=64)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream
lib = _lib.load()
H, HD = 16, 64
C = H * HD
# Sky-16f train step (B = 6); UCF-128f revise forward and the first sampler step (B = 4); Taichi-16f sampling (B = 16)
SHAPES = [(6, "c2 enc", 256, 512), (6, "c2 self", 256, 256), (6, "c2 dec", 512, 256), (6, "c2 lt2l", 256, 768),
          (4, "c4 enc revise", 256, 7936), (4, "c4 lt2l revise", 256, 512), (4, "c4 dec sample0", 8192, 256), (4, "c4 lt2l sample0", 256, 8448),
          (4, "c4 train enc", 256, 4096), (4, "c4 train dec", 4096, 256), (16, "c5 enc", 256, 512), (16, "c5 self", 256, 256)]
for B, name, NQ, NK in SHAPES:
    q = torch.randn(B, NQ, C, device="cuda").bfloat16()
    kv = torch.randn(B, NK, 2 * C, device="cuda").bfloat16()
    do = torch.randn(B, NQ, C, device="cuda").bfloat16()
    o = torch.empty_like(q); dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    lse = torch.empty(B, H, NQ, device="cuda"); delta = torch.empty(B, H, NQ, device="cuda")
    vp = kv.data_ptr() + C * 2; dvp = dkv.data_ptr() + C * 2
    fwd = lambda: check(lib.mebt_op_attention_fwd(1, ptr(q), ptr(kv), vp, ptr(o), ptr(lse), B, H, NQ, NK, HD, C, 2 * C, 2 * C, C, 0, cur_stream()))
    bwd = lambda: check(lib.mebt_op_attention_bwd(1, ptr(q), ptr(kv), vp, ptr(o), ptr(lse), ptr(do), ptr(dq), ptr(dkv), dvp, ptr(delta), B, H, NQ, NK, HD, C, 2 * C, 2 * C, C, 0, cur_stream()))
    for fn, lab, fl in ((fwd, "fwd", 4.0 * B * H * NQ * NK * HD), (bwd, "bwd(dq+dkv)", 14.0 * B * H * NQ * NK * HD)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{name:16s} B={B:2d} {lab:12s} NQ={NQ} NK={NK}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s")
