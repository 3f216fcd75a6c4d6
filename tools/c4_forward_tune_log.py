#!/usr/bin/env python3
"""Config 4 (UCF-101 128f, block 8192): one revise pass at (NC, NT) = (7936, 256), batch 4, with the tuner's decisions printed
(MEBT_GEMM_TUNE_LOG=1) and the forward timed.  GPU box: MEBT_GEMM_TUNE_LOG=1 python tools/c4_forward_tune_log.py"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import presets

ucfg = presets.ucf_128f()
torch.manual_seed(1)
um = presets.build_model(ucfg, compute_dtype="bf16").cuda().eval()
shape = tuple(ucfg.model.mask.params.shape)
xr = torch.randint(0, 16384, (4, *shape), device="cuda")
with torch.no_grad():
    f = lambda: um.draft_and_revise(xr, None, 8, 1.0, None, None, 2, 1.0, None, None, 32, True)
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"64 revise forwards: {dt:.3f} s ({dt / 64 * 1e3:.2f} ms per forward)")
