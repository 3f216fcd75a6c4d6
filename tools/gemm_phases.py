#!/usr/bin/env python3
"""Where does a small GEMM's launch time go?  In-kernel s_memtime stamps (mebt_debug_gemm_stamps) of every workgroup of one
bf16 GEMM — entry, first k-tile landed, main loop done, epilogue retired — cold (operands flushed from L2 / Infinity Cache) and
warm, next to the HIP-event time of the launch.  s_memtime ticks at the shader clock; converted with the 100 MHz
s_memrealtime ratio measured by a tiny calibration kernel is not needed here: we report cycles and the event time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib
from mebt_amd._lib import check, ptr, cur_stream

lib = _lib.load()
dev = "cuda"
shapes = [(1536, 1024, 1024, 96, 64), (1536, 1024, 4096, 96, 64), (1536, 4096, 1024, 64, 128), (3072, 1024, 1024, 96, 128)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for M, N, K, bm, bn in shapes:
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    nwg = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    stamps = torch.zeros(nwg * 4, dtype=torch.int64, device=dev)
    lib.mebt_debug_gemm_tile(bm, bn)
    lib.mebt_debug_gemm_variant(3)            # LDS-DMA ring of 3, one pipeline: the stamped kernel
    for mode in ("cold", "warm"):
        res = []
        for rep in range(6):
            if mode == "cold":
                flush.fill_(rep)
            lib.mebt_debug_gemm_stamps(ptr(stamps))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.mebt_op_gemm(_lib.BF16, ptr(A), ptr(B), ptr(Cc), None, None, None, M, N, K, K, K, N, N, 1, 1, 0, 0, 0, 1, cur_stream()))
            e1.record()
            torch.cuda.synchronize()
            lib.mebt_debug_gemm_stamps(None)
            s = stamps.view(nwg, 4).cpu().double()
            t0 = s[:, 0].min()
            res.append((e0.elapsed_time(e1) * 1e3, (s[:, 0] - t0).median().item(), (s[:, 1] - s[:, 0]).median().item(), (s[:, 2] - s[:, 1]).median().item(),
                        (s[:, 3] - s[:, 2]).median().item(), (s[:, 3] - t0).max().item(), (s[:, 0] - t0).max().item()))
        r = torch.tensor(res[2:]).median(0).values
        print(f"{M}x{N}x{K} tile {bm}x{bn} {mode}: event {r[0]:.1f} us | cycles: start skew (median/max) {r[1]:.0f}/{r[6]:.0f}, first tile {r[2]:.0f}, "
              f"main loop {r[3]:.0f}, epilogue {r[4]:.0f}, last workgroup done at {r[5]:.0f}")
    lib.mebt_debug_gemm_tile(0, 0)
    lib.mebt_debug_gemm_variant(-1)
