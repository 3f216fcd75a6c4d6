#!/usr/bin/env python3
"""3D-VQGAN first stage at the BASELINE config-5 geometry ([B,3,16,128,128] <-> [B,4,16,16], 16384 codes x 256), random-init
weights: wall time of encode / decode per dtype and batch (rocprofv3 --kernel-trace --stats on this script gives the kernels)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import presets
from mebt_amd.vqgan import VQGAN

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
m = VQGAN(presets.vqgan_args()).cuda().eval()
vid = torch.rand(B, 3, 16, 128, 128, device="cuda") - 0.5
for dtype in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("f16", "f32")):
    m.compute_dtype = dtype
    ids = m.encode(vid)
    rec = m.decode(ids)
    torch.cuda.synchronize()
    n = 5 if dtype == "f16" else 2
    t0 = time.perf_counter()
    for _ in range(n):
        ids = m.encode(vid)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        rec = m.decode(ids)
    torch.cuda.synchronize()
    td = (time.perf_counter() - t0) / n
    print(f"{dtype} B={B}: encode {te * 1e3:.2f} ms ({B * 2 * 23.83e9 / te / 1e12:.1f} TFLOP/s)  decode {td * 1e3:.2f} ms ({B * 2 * 347.17e9 / td / 1e12:.1f} TFLOP/s)")
