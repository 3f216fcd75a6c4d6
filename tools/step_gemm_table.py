#!/usr/bin/env python3
"""Per-signature table of the GEMM-family launches INSIDE the Sky-16f train step (HIP events on the launch stream, tuned
configurations, the operands where the step leaves them: L2-cold activations, weights prefetched into the Infinity Cache), next to
torch.matmul (hipBLASLt / rocBLAS) on the same shapes with HBM-cold operands.  tools/gemm_bench.py times the operator-level calls,
which use the heuristic tile — this is the table of what the step really launches (profiles/r05_gemm_in_step_vs_hipblaslt.txt).
Usage (GPU box): python tools/step_gemm_table.py [--steps 6] [--no-torch]"""
import argparse
import collections
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import _lib, presets          # noqa: E402
from mebt_amd.trainer import TrainLoop      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--no-torch", action="store_true")
ap.add_argument("--batch", type=int, default=6)
ap.add_argument("--t", type=float, default=0.5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
lib = _lib.load()
cfg = presets.sky_16f(vtokens=True, dropout=0.1)
torch.manual_seed(0)
model = presets.build_model(cfg, compute_dtype="bf16").to(dev).train()
loop = TrainLoop(model)
g = torch.Generator().manual_seed(1234)
shape = cfg.model.mask.params.shape
N = shape[0] * shape[1] * shape[2]
x = torch.randint(0, 16384, (args.batch, *shape), generator=g).to(dev)
idx = torch.stack([torch.randperm(N, generator=g) for _ in range(args.batch)]).to(dev)
for _ in range(4):
    loop.step(x, idx, t=args.t)
torch.cuda.synchronize()
lib.mebt_profile_enable(1)
for _ in range(args.steps):
    loop.step(x, idx, t=args.t)
torch.cuda.synchronize()
n = lib.mebt_profile_dump(None, 0)
buf = C.create_string_buffer(int(n))
lib.mebt_profile_dump(buf, n)
lib.mebt_profile_enable(0)
rows = collections.OrderedDict()
for line in buf.value.decode().splitlines():
    f = line.split()
    key = tuple([f[0]] + [int(v) for v in f[1:8]])
    r = rows.setdefault(key, [0, 0.0, 0.0])
    r[0] += 1
    r[1] += float(f[8])
    r[2] += float(f[9])
EPI = {0: "", 1: "+gelu", 2: "+resid", 3: "+gelu'", 4: "+adamw"}


def torch_us(M, Nn, K, akc, bkc):
    """torch.matmul on HBM-cold operands of the same layouts (rotating through a > 600 MB pool)"""
    pool = max(2, int(6e8 // ((M * K + Nn * K) * 2)))
    A = torch.randn((pool,) + ((M, K) if akc else (K, M)), device=dev).bfloat16()
    B = torch.randn((pool,) + ((Nn, K) if bkc else (K, Nn)), device=dev).bfloat16()
    out = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)

    def run(i):
        a = A[i % pool] if akc else A[i % pool].t()
        b = B[i % pool].t() if bkc else B[i % pool]
        torch.matmul(a, b, out=out)
    for i in range(3):
        run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        run(3 + i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / 20


print(f"# in-step GEMM-family launches, Sky-16f train step (B = {args.batch}, t = {args.t}), HIP events, {args.steps} steps; torch = torch.matmul, HBM-cold operands")
print(f"{'launch':58s} {'per step':>8s} {'us':>8s} {'TF/s':>7s} {'ms/step':>8s} | {'torch us':>9s} {'ms/step lost':>12s}")
tot = lost = tot_t = 0.0
for key, (cnt, ms, gf) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    per = cnt / args.steps
    us = 1e3 * ms / cnt
    tf = gf / ms if ms > 0 else 0.0
    tag = key[0]
    if tag == "g":
        M, Nn, K, akc, bkc, epi, cf = key[1:8]
        name = f"{'fwd' if bkc else 'dgrad'} {M}x{Nn}x{K}{EPI.get(epi, '')}{' f32out' if cf else ''}" if akc else f"wgrad {M}x{Nn}x{K}"
        t_us = None if args.no_torch else torch_us(M, Nn, K, akc, bkc)
    elif tag == "p":
        M0, N0, K0, M1, N1, K1, bkc = key[1:8]
        name = f"pair {'fwd' if bkc else 'dgrad'} {M0}x{N0}x{K0} + {M1}x{N1}x{K1}"
        t_us = None if args.no_torch else torch_us(M0, N0, K0, 1, bkc) + torch_us(M1, N1, K1, 1, bkc)
    else:
        name = f"grouped wgrad {key[1]} items, {key[2]} Ki outputs, K <= {key[3]} (+AdamW)"
        t_us = None
    tot += ms / args.steps
    line = f"{name:58s} {per:8.1f} {us:8.1f} {tf:7.0f} {ms / args.steps:8.3f} |"
    if t_us is not None:
        d = (us - t_us) * per * 1e-3
        lost += max(0.0, d)
        tot_t += min(us, t_us) * per * 1e-3
        line += f" {t_us:9.1f} {d:12.3f}"
    else:
        tot_t += ms / args.steps
    print(line)
print(f"# GEMM family: {tot:.3f} ms per step by events; rows slower than torch.matmul lose {lost:.3f} ms per step in all; best-of-both {tot_t:.3f} ms")
