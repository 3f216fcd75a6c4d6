#!/bin/bash
# In-step refinement of the shipped GEMM table on an MI355X (round 6): tools/step_tune.py from a fresh in-situ tuning, its choices merged
# over mebt_amd/tune/gfx950.txt, then an alternating A/B of the shipped table against the merged one (bench.py, ms per step).
#   tools/refine_shipped_table.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/tune
R=${1:-3}
MEBT_GEMM_TUNE_SHIPPED=0 python3 tools/step_tune.py gpurun_out/tune/refined_r6.txt gpurun_out/tune/step_tune_r6.log 2> gpurun_out/tune/step_tune_r6.err
tail -25 gpurun_out/tune/step_tune_r6.log
python3 - <<'PY'
import os
root = os.environ.get("GRAFT_REPO_ROOT", ".")
def read(p):
    lines = open(p).read().splitlines()
    return lines[0], dict(l.rsplit(" ", 1) for l in lines[1:] if l.strip())
ver, table = read(os.path.join(root, "mebt_amd/tune/gfx950.txt"))
ver2, refined = read(os.path.join(root, "gpurun_out/tune/refined_r6.txt"))
assert ver == ver2
n = sum(1 for k, v in refined.items() if table.get(k) != v)
table.update(refined)
with open(os.path.join(root, "gpurun_out/tune/gfx950_r6_refined.txt"), "w") as f:
    f.write(ver + "\n" + "".join(f"{k} {v}\n" for k, v in table.items()))
print(f"{len(refined)} refined signatures merged ({n} differ from the shipped table); {len(table)} entries")
PY
run() { python3 bench.py --steps 30 --warmup 5 --windows 1 --secondary none --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['gemm_ms_per_step'])"; }
for r in $(seq 1 $R); do
  a=$(run)
  cp gpurun_out/tune/gfx950_r6_refined.txt /tmp/refined_cache.txt
  b=$(MEBT_GEMM_TUNE_SHIPPED=0 MEBT_GEMM_TUNE_CACHE=/tmp/refined_cache.txt run)
  echo "  [shipped] $a   [shipped + in-step refinement] $b"
done
