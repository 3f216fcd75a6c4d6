#!/bin/bash
# rocprofv3 kernel statistics of EACH leg of BASELINE config 4 on its own (revise / sample / bootstrap / train), so that a leg's time in
# the bench line can be set against its own kernel table:   tools/profile_c4_legs.sh <tag>
TAG=${1:-c4legs}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
for LEG in revise sample bootstrap train; do
  ARGS="$ROOT/bench.py --secondary c4 --c4-legs $LEG --no-cpu-baseline"
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${LEG}_stats" -o bench -- python3 $ARGS > "$OUT/c4_${LEG}_run.json" 2> "$OUT/c4_${LEG}.err"
  cp "$OUT/${LEG}_stats/"*kernel_stats.csv "$OUT/c4_${LEG}_kernel_stats.csv" 2>/dev/null
  rm -rf "$OUT/${LEG}_stats"
  cd "$ROOT" && python3 tools/kernel_table.py "$OUT/c4_${LEG}_kernel_stats.csv" 14 > "$OUT/c4_${LEG}_kernel_table.txt" 2>&1
  echo "== $LEG"; head -c 400 "$OUT/c4_${LEG}_run.json"; echo; tail -12 "$OUT/c4_${LEG}_kernel_table.txt"
done
