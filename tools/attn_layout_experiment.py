#!/usr/bin/env python3
"""Long-key forward attention (config 4's latent_enc: 256 latent queries x 7936 keys, B = 4, 16 heads): does the kernel's time
depend on the key / value row pitch?  (a) rows of all heads interleaved, pitch 4 KiB (the K/V projection's output and the sampling
loops' cache); (b) one contiguous [NK, 64] slab per (sample, head), pitch 128 B — the same launch geometry, expressed as B' = B x H
samples of one head.  Buffers rotate through POOL sets so that keys / values come from HBM as in a forward (7 layers x 134 MB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream
lib = _lib.load()
H, HD, B, NQ = 16, 64, 4, 256
C = H * HD
POOL = 5
for NK in (7936, 4096, 1024):
    for pool in (1, POOL):
        qa = torch.randn(B, NQ, C, device="cuda").bfloat16(); oa = torch.empty_like(qa)
        kva = [torch.randn(B, NK, 2 * C, device="cuda").bfloat16() for _ in range(pool)]
        qb = torch.randn(B * H, NQ, HD, device="cuda").bfloat16(); ob = torch.empty_like(qb)
        kb = [torch.randn(B * H, NK, HD, device="cuda").bfloat16() for _ in range(pool)]
        vb = [torch.randn(B * H, NK, HD, device="cuda").bfloat16() for _ in range(pool)]
        lse = torch.empty(B, H, NQ, device="cuda")
        fa = lambda i: check(lib.mebt_op_attention_fwd(1, ptr(qa), ptr(kva[i % pool]), kva[i % pool].data_ptr() + C * 2, ptr(oa), ptr(lse), B, H, NQ, NK, HD, C, 2 * C, 2 * C, C, 0, cur_stream()))
        fb = lambda i: check(lib.mebt_op_attention_fwd(1, ptr(qb), ptr(kb[i % pool]), ptr(vb[i % pool]), ptr(ob), ptr(lse), B * H, 1, NQ, NK, HD, HD, HD, HD, HD, 0, cur_stream()))
        if pool == 1:       # the launch against fp32 softmax(q k^T / 8) v
            fa(0); torch.cuda.synchronize()
            k = kva[0][..., :C].float().view(B, NK, H, HD).transpose(1, 2); v = kva[0][..., C:].float().view(B, NK, H, HD).transpose(1, 2)
            ref = torch.softmax(qa.float().view(B, NQ, H, HD).transpose(1, 2) @ k.transpose(-1, -2) / 8.0, -1) @ v
            print(f"NK={NK:5d} max |o - ref| = {(oa.float().view(B, NQ, H, HD).transpose(1, 2) - ref).abs().max().item():.2e}")
        for fn, lab in ((fa, "interleaved heads, pitch 4096 B"), (fb, "per-head slabs,    pitch  128 B")):
            for i in range(5): fn(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(40): fn(i)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 40
            fl = 4.0 * B * H * NQ * NK * HD
            print(f"NK={NK:5d} {'HBM-cold' if pool > 1 else 'cache-warm'}  {lab}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s  {B * NK * 2 * C * 2 / us / 1e6:5.2f} TB/s of keys+values")
