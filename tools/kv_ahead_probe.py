#!/usr/bin/env python3
"""Does the K/V-ahead side stream really run beside the latent chain?  Config 4's revise forward (B = 4, NC = 7936, NT = 256) timed
with MEBT_KV_AHEAD = 0 / 1 on the model's own (lowest-priority) side stream and on a probed torch stream, plus the serialisation
probe of parallel.streams_serialised for that stream.   python tools/kv_ahead_probe.py [B] [NC] [NT]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import presets
from mebt_amd.parallel import pick_concurrent_stream, streams_serialised
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 7936
NT = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda", 0)
cfg = presets.ucf_128f()


def run(tag, mode, probed):
    os.environ["MEBT_KV_AHEAD"] = mode
    torch.manual_seed(0)
    m = presets.build_model(cfg, compute_dtype="bf16").to(dev).eval()
    N = 8192
    x = torch.randint(0, 16384, (B, N), device=dev)
    perm = torch.stack([torch.randperm(N, device=dev) for _ in range(B)])
    ci, ti = perm[:, :NC].contiguous(), perm[:, NC:NC + NT].contiguous()
    with torch.no_grad():
        m.reconstruct_mask(x, ci, ti)
        if probed:
            s = pick_concurrent_stream(torch.cuda.current_stream(), priority=probed_priority)
            m._native.lib.mebt_debug_set_side_stream(m._native.h, s.cuda_stream)
            ser = (streams_serialised(torch.cuda.current_stream(), s), streams_serialised(s, torch.cuda.current_stream()))
        else:
            ser = None
        for _ in range(3):
            m.reconstruct_mask(x, ci, ti)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                m.reconstruct_mask(x, ci, ti)
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
    print(f"{tag:44s} {best:7.3f} ms per forward   serialised(main->side, side->main) = {ser}", flush=True)
    del m
    torch.cuda.empty_cache()


probed_priority = 0
run("in line (MEBT_KV_AHEAD=0)", "0", False)
run("ahead, model's own lowest-priority stream", "1", False)
run("ahead, probed torch stream (priority 0)", "1", True)
probed_priority = -1
run("ahead, probed torch stream (priority -1)", "1", True)
run("in line (MEBT_KV_AHEAD=0)", "0", False)
