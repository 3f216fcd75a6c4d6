#!/bin/bash
OUT=gpurun_out/r4k; mkdir -p $OUT
export MEBT_GEMM_TUNE_CACHE=$PWD/$OUT/tune.txt
for i in 1 2; do
for D in 0.1 0.0; do python bench.py --dropout $D --secondary none --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('dropout $D', r['ms_per_step'], r['roofline']['gemm_ms_per_step'])"; done
done
python - <<'PY'
import torch, time
from mebt_amd import presets
from mebt_amd.trainer import TrainLoop
# per-family event timing is not exposed: time the step with attention dropout only / residual dropout only
for name, kw in (("attn only", dict(attn=0.1, resid=0.0, embd=0.0)), ("resid+embd only", dict(attn=0.0, resid=0.1, embd=0.1))):
    cfg = presets.sky_16f(vtokens=True, dropout=0.1)
    p = cfg.model.params
    p.attn_pdrop, p.resid_pdrop, p.embd_pdrop = kw["attn"], kw["resid"], kw["embd"]
    torch.manual_seed(0)
    m = presets.build_model(cfg, compute_dtype="bf16").cuda().train()
    loop = TrainLoop(m)
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 16384, (6, 4, 16, 16), generator=g).cuda()
    idx = torch.stack([torch.randperm(1024, generator=g) for _ in range(6)]).cuda()
    for _ in range(6): loop.step(x, idx, t=0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): loop.step(x, idx, t=0.5)
    torch.cuda.synchronize(); print(name, round((time.perf_counter() - t0) / 30 * 1e3, 3), "ms")
    del m, loop
PY
