#!/usr/bin/env python3
"""Wall time of the samplers on the Sky-16f network (random weights): MaskGIT `sample` and `draft_and_revise`
(reference transformer.py:353-447, 632-663), batch 4, 1024 tokens."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import presets
cfg = presets.sky_16f(vtokens=True, dropout=0.0)
torch.manual_seed(0)
model = presets.build_model(cfg, compute_dtype="bf16").to("cuda").eval()
B = 4
x = torch.zeros(B, 4, 16, 16, dtype=torch.long, device="cuda")
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
with torch.no_grad():
    t = timed(lambda: model.sample(x, None, 1.0, None, None, 16, None, None, context_temperature=4.5, skips=False))
    print(f"sample, 16 steps, B={B}: {t * 1e3:.1f} ms  ({B * 1024 / t:.0f} tokens/s)")
    t = timed(lambda: model.sample(x, None, 1.0, 32, None, 16, None, None, context_temperature=4.5, skips=False))
    print(f"sample top-k 32 (the shipped scripts' setting), 16 steps: {t * 1e3:.1f} ms")
    t = timed(lambda: model.sample(x, None, 1.0, 64, 0.95, 16, None, None, context_temperature=4.5, skips=False))
    print(f"sample top-k 64 / top-p 0.95, 16 steps: {t * 1e3:.1f} ms")
    x0 = torch.randint(0, 16384, (B, 4, 16, 16), device="cuda")
    t = timed(lambda: model.draft_and_revise(x0, None, 8, 1.0, None, None, 8, 1.0, None, None, 2, False), n=2)
    print(f"draft_and_revise (8 draft, 8 revise x2): {t * 1e3:.1f} ms")
