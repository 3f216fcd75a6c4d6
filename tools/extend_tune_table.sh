#!/bin/bash
# ADD the signatures the shipped table (mebt_amd/tune/gfx950.txt) does not hold yet, without touching its entries (they carry the
# in-step refinement of tools/make_tune_table.sh):  tools/extend_tune_table.sh
# The shipped table stays loaded; the in-situ tuner only sees what it misses (new kernels paths of a round: bf16 logits of the
# sampling loops, the key / value cache's dirty-row projections) and appends it to a cache file, which is merged into the table.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/tune
export MEBT_GEMM_TUNE_CACHE=$ROOT/gpurun_out/tune/added.txt
rm -f $MEBT_GEMM_TUNE_CACHE
python3 bench.py --no-cpu-baseline > gpurun_out/tune/ext_full.json 2> gpurun_out/tune/ext_full.err
python3 bench.py --secondary c5 --c5-batch 4 --no-cpu-baseline > gpurun_out/tune/ext_c5b4.json 2>> gpurun_out/tune/ext_full.err
MEBT_SAMPLE_BF16_LOGITS=0 python3 bench.py --secondary c4 --c4-legs revise,sample,bootstrap --no-cpu-baseline > gpurun_out/tune/ext_c4_f32logits.json 2>> gpurun_out/tune/ext_full.err
# the data-parallel step (weight gradients stored as bf16 / fp32 instead of consumed by the fused AdamW epilogue: other grouped signatures)
MEBT_DP_FORCE=1 python3 bench.py --secondary none --no-cpu-baseline --steps 5 > gpurun_out/tune/ext_dp.json 2>> gpurun_out/tune/ext_full.err
MEBT_DP_FORCE=1 MEBT_DP_MODE=allreduce python3 bench.py --secondary none --no-cpu-baseline --steps 5 > gpurun_out/tune/ext_dp_ar.json 2>> gpurun_out/tune/ext_full.err
python3 - <<'PY'
import os
root = os.environ.get("GRAFT_REPO_ROOT", ".")
def read(p):
    lines = open(p).read().splitlines()
    return lines[0], dict(l.rsplit(" ", 1) for l in lines[1:] if l.strip())
ver, table = read(os.path.join(root, "mebt_amd/tune/gfx950.txt"))
ver2, added = read(os.path.join(root, "gpurun_out/tune/added.txt"))
assert ver == ver2, (ver, ver2)
new = {k: v for k, v in added.items() if k not in table}
table.update(new)
out = os.path.join(root, "gpurun_out/tune/gfx950_extended.txt")
with open(out, "w") as f:
    f.write(ver + "\n" + "".join(f"{k} {v}\n" for k, v in table.items()))
print(f"{len(new)} signatures added to {len(table) - len(new)}; written to {out}")
PY
head -c 300 gpurun_out/tune/ext_full.json
