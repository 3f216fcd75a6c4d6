#!/bin/bash
# Regenerate the shipped table of tuned GEMM configurations (mebt_amd/tune/gfx950.txt) on an MI355X:  tools/make_tune_table.sh
# Starts from an EMPTY table (MEBT_GEMM_TUNE_SHIPPED=0) and lets the in-situ tuner see every signature of the shipped configs:
# the full bench (config 2: fused and separate optimizer, t sweep, the 100-step t ~ U(0,1) run = every 128-token bucket, samplers;
# config 4: revise / sample / bootstrap / train step; config 5 at batch 16), then the data-parallel step (bf16 wire gradients).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/tune
export MEBT_GEMM_TUNE_SHIPPED=0 MEBT_GEMM_TUNE_CACHE=$ROOT/gpurun_out/tune/gfx950.txt
rm -f $MEBT_GEMM_TUNE_CACHE
python3 bench.py --no-cpu-baseline > gpurun_out/tune/bench_full.json 2> gpurun_out/tune/bench_full.err
MEBT_DP_FORCE=1 python3 bench.py --secondary none --no-cpu-baseline --steps 5 > gpurun_out/tune/bench_dp.json 2> gpurun_out/tune/bench_dp.err
MEBT_DP_FORCE=1 MEBT_DP_MODE=allreduce python3 bench.py --secondary none --no-cpu-baseline --steps 5 > gpurun_out/tune/bench_dp_ar.json 2>> gpurun_out/tune/bench_dp.err
# in-step refinement of the benchmarked step's signatures (tools/step_tune.py: a fresh in-process tuning, then the runner-ups of every
# signature timed inside the step; measured -0.09 ms per step in a cross-process A/B, profiles/r04_step_tune_ab.txt); its choices
# replace the table's entries of the same signatures
MEBT_GEMM_TUNE_CACHE= python3 tools/step_tune.py gpurun_out/tune/refined.txt gpurun_out/tune/step_tune.log 2> gpurun_out/tune/step_tune.err
python3 - <<'PY'
import os
root = os.environ.get("GRAFT_REPO_ROOT", ".")
path = os.path.join(root, "gpurun_out/tune/gfx950.txt")
def read(p):
    lines = open(p).read().splitlines()
    return lines[0], dict(l.rsplit(" ", 1) for l in lines[1:] if l.strip())
ver, table = read(path)
ver2, refined = read(os.path.join(root, "gpurun_out/tune/refined.txt"))
assert ver == ver2
n = sum(1 for k, v in refined.items() if table.get(k) != v)
table.update(refined)
with open(path, "w") as f:
    f.write(ver + "\n" + "".join(f"{k} {v}\n" for k, v in table.items()))
print(f"{len(refined)} refined signatures merged ({n} differ from the isolated pick); {len(table)} entries")
PY
cat gpurun_out/tune/step_tune.log; wc -l $MEBT_GEMM_TUNE_CACHE; head -c 400 gpurun_out/tune/bench_full.json
