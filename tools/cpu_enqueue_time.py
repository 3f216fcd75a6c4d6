#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one train step (no GPU sync) vs the GPU time of the step?
If the two are close the step is launch-bound."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mebt_amd import presets
from mebt_amd.parallel import GradReducer
from mebt_amd.trainer import TrainLoop
import bench
cfg = presets.sky_16f(vtokens=True, dropout=0.1)
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = presets.build_model(cfg, compute_dtype="bf16").to(dev).train()
loop = TrainLoop(model, GradReducer(world_size=1))
x, idx = bench.synthetic_batch(6, cfg.model.mask.params.shape, 0, dev)
for _ in range(5): loop.step(x, idx, t=0.5)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10): loop.step(x, idx, t=0.5)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0) / 10:.2f} ms/step, total {1e3 * (t2 - t0) / 10:.2f} ms/step")
# single step from an idle GPU: enqueue latency matters here
torch.cuda.synchronize()
t0 = time.perf_counter(); loop.step(x, idx, t=0.5); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"one step from idle: enqueue {1e3 * (t1 - t0):.2f} ms, total {1e3 * (t2 - t0):.2f} ms")
