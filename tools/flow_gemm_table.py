#!/usr/bin/env python3
"""Per-signature table of the GEMM-family launches of an inference flow at config 4 geometry (block 8192, batch 4): a short
`bidirect_sample` (bootstrap + MaskGIT steps) or a revise pass, HIP events on the launch stream (mebt_profile_dump).
Usage (GPU box): python tools/flow_gemm_table.py [bootstrap|revise]"""
import collections
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib, presets          # noqa: E402
from mebt_amd.sampling import bidirect_sample          # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "bootstrap"
dev = torch.device("cuda", 0)
lib = _lib.load()
cfg = presets.ucf_128f()
torch.manual_seed(0)
m = presets.build_model(cfg, compute_dtype="bf16").to(dev).eval()
m.mask_sampler.schedule = "cosine"


def flow():
    with torch.no_grad():
        if what == "bootstrap":
            bidirect_sample(m, 4, 128, 128, 128, temperature=1.0, top_k=32, top_p=None, vid_n_steps=8, vid_c_temp=2.0, bootstrap=8)
        else:
            x = torch.randint(0, 16384, (4, 32, 16, 16), device=dev)
            m.draft_and_revise(x, None, 8, 1.0, None, None, 32, 1.0, None, None, 1, True)


flow()
torch.cuda.synchronize()
lib.mebt_profile_enable(1)
flow()
torch.cuda.synchronize()
n = lib.mebt_profile_dump(None, 0)
buf = C.create_string_buffer(int(n))
lib.mebt_profile_dump(buf, n)
lib.mebt_profile_enable(0)
rows = collections.OrderedDict()
for line in buf.value.decode().splitlines():
    f = line.split()
    key = tuple([f[0]] + [int(v) for v in f[1:8]])
    if key[0] == "g":          # bucket the token-count dimension like the tuner does
        key = (key[0], (key[1] + 127) // 128 * 128 if key[1] > 128 else key[1]) + key[2:]
    r = rows.setdefault(key, [0, 0.0, 0.0])
    r[0] += 1
    r[1] += float(f[8])
    r[2] += float(f[9])
tot = sum(r[1] for r in rows.values())
print(f"# {what}: {sum(r[0] for r in rows.values())} GEMM-family launches, {tot:.1f} ms by events")
for key, (cnt, ms, gf) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{str(key):64s} x{cnt:5d} {1e3 * ms / cnt:9.1f} us {gf / ms if ms else 0:7.0f} TF/s {100 * ms / tot:5.1f} %")
