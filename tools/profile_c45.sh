#!/bin/bash
# rocprofv3 kernel statistics of BASELINE configs 4 and 5:  tools/profile_c45.sh <tag>
# `bench.py --secondary c4|c5` runs ONLY that configuration's legs (no headline step), python3 directly after `--`.
# Each leg runs twice inside the process (one untimed call that also tunes unseen GEMM signatures, one timed), so per-kernel
# averages are over both; the tune cache is populated by an unprofiled run first so that no tuning candidates are traced.
TAG=${1:-c45}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export MEBT_GEMM_TUNE_CACHE=$OUT/tune_cache.txt
for CFG in c4 c5; do
  ARGS="$ROOT/bench.py --secondary $CFG --no-cpu-baseline"
  cd "$ROOT" && python3 $ARGS > "$OUT/${CFG}_populate.json" 2> "$OUT/${CFG}_populate.err"
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${CFG}_stats" -o bench -- python3 $ARGS > "$OUT/${CFG}_stats.json" 2> "$OUT/${CFG}_stats.err"
  cp "$OUT/${CFG}_stats/"*kernel_stats.csv "$OUT/${CFG}_kernel_stats.csv" 2>/dev/null
  rm -rf "$OUT/${CFG}_stats"
  cd "$ROOT" && python3 tools/kernel_table.py "$OUT/${CFG}_kernel_stats.csv" 25 > "$OUT/${CFG}_kernel_table.txt" 2>&1
  tail -30 "$OUT/${CFG}_kernel_table.txt"
done
