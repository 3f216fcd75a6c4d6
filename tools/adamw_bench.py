#!/usr/bin/env python3
"""AdamW kernel time on the Sky-16f parameter set (env MEBT_ADAMW_NT / MEBT_ADAMW_BLOCKS select variants)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import presets
cfg = presets.sky_16f(vtokens=True, dropout=0.0)
model = presets.build_model(cfg, compute_dtype="bf16").to("cuda").train()
nm = model._ensure_native()
nm.ensure_grads()
for _ in range(3): nm.adamw_step(1e-4, 0.01, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(10): nm.adamw_step(1e-4, 0.01, 2 + i)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
n = nm.W.numel() + nm.P.numel()
print(f"NT={os.environ.get('MEBT_ADAMW_NT','0')} blocks={os.environ.get('MEBT_ADAMW_BLOCKS','4096')}: {ms*1e3:.0f} us/step, {n*30/ms/1e9:.2f} TB/s")
