#!/usr/bin/env python3
"""In-step refinement of the GEMM tuner's choices at the benchmarked config (Sky-16f, batch 6, t = 0.5, fused optimizer).

The in-situ tuner ranks the candidates of a signature by their ISOLATED time (cold L2, warm Infinity Cache).  In the step a product
runs between its real neighbours: the predecessor's weight prefetch, the successor's ramp, the other workgroups' tails.  This tool
starts from a fresh in-situ tuning (MEBT_GEMM_TUNE_SHIPPED=0 recommended), then walks the signatures in the order of their time
share and times the WHOLE step with each of the tuner's runner-up candidates (mebt_gemm_tune_alternatives) in place of its pick,
keeping a change only if the step is faster in two independent measurements.  Output: the refined table (text) + a log.

  python tools/step_tune.py out_table.txt [log.txt]
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib, presets
from mebt_amd.trainer import TrainLoop

out_path = sys.argv[1]
log = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stderr
lib = _lib.load()
cfg = presets.sky_16f(vtokens=True, dropout=0.1)
torch.manual_seed(0)
model = presets.build_model(cfg, compute_dtype="bf16").cuda().train()
loop = TrainLoop(model)
g = torch.Generator().manual_seed(1234)
x = torch.randint(0, 16384, (6, 4, 16, 16), generator=g).cuda()
idx = torch.stack([torch.randperm(1024, generator=g) for _ in range(6)]).cuda()
for _ in range(4):
    loop.step(x, idx, t=0.5)
torch.cuda.synchronize()


def step_ms(n=30):
    for _ in range(3):
        loop.step(x, idx, t=0.5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        loop.step(x, idx, t=0.5)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def alternatives():
    n = lib.mebt_gemm_tune_alternatives(None, 0)
    buf = C.create_string_buffer(int(n))
    lib.mebt_gemm_tune_alternatives(buf, n)
    out = []
    for line in buf.value.decode().splitlines():
        key, vals = line.split(" : ")
        v = vals.split()
        out.append((key, [(int(v[i]), float(v[i + 1])) for i in range(0, len(v), 2)]))
    return out


with open(out_path + ".before", "w") as f:
    f.write(_lib.tune_table_text())
version = _lib.tune_table_text().splitlines()[0]
table = dict(l.rsplit(" ", 1) for l in _lib.tune_table_text().splitlines()[1:])
alts = [(k, v) for k, v in alternatives() if k in table and len(v) > 1]
alts.sort(key=lambda kv: -kv[1][0][1])                  # slowest (largest share) signatures first
base = min(step_ms(), step_ms())
print(f"baseline {base:.3f} ms per step; {len(alts)} signatures with runner-ups", file=log, flush=True)
changed = 0
for rnd in range(2):                       # a second round: the neighbours have changed
  for key, cands in alts:
    cur = int(table[key])
    for val, us in cands[:4]:
        if val == cur:
            continue
        _lib.tune_table_merge(f"{version}\n{key} {val}\n", overwrite=True)
        t1 = step_ms()
        if t1 < base - 0.012:
            t2 = step_ms()
            if t2 < base - 0.012:
                print(f"  round {rnd + 1}  {key}: {cur} -> {val} (isolated {us:.1f} us): step {base:.3f} -> {max(t1, t2):.3f} ms", file=log, flush=True)
                base, cur, changed = max(t1, t2), val, changed + 1
                table[key] = str(val)
                continue
        _lib.tune_table_merge(f"{version}\n{key} {cur}\n", overwrite=True)
final = min(step_ms(), step_ms())
print(f"{changed} choices changed; step now {final:.3f} ms", file=log, flush=True)
with open(out_path, "w") as f:
    f.write(_lib.tune_table_text())
