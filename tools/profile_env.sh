#!/bin/bash
# Kernel trace + per-queue timeline of the train step under an environment setting:  tools/profile_env.sh <tag> [NAME=value ...]
# (first a run that fills a tune cache of its own, then rocprofv3 --kernel-trace --stats of the same command; tools/kernel_table.py and
#  tools/trace_timeline.py print the per-kernel table and how busy every hardware queue was)
TAG=${1:-env}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
export MEBT_GEMM_TUNE_CACHE=$OUT/tune_cache.txt
ARGS="$ROOT/bench.py --steps 10 --warmup 5 --secondary none --no-cpu-baseline"
cd "$ROOT" && python3 $ARGS > "$OUT/populate.json" 2> "$OUT/populate.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 $ARGS > "$OUT/stats.json" 2> "$OUT/stats.err"
cd "$ROOT"
cp "$OUT/stats/"*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
python3 tools/kernel_table.py "$OUT/kernel_stats.csv" 40 > "$OUT/kernel_table.txt"
python3 tools/trace_timeline.py "$OUT/stats/"*kernel_trace.csv > "$OUT/timeline.txt" 2>&1
rm -rf "$OUT/stats"
head -c 300 "$OUT/stats.json"; echo; cat "$OUT/kernel_table.txt"; head -40 "$OUT/timeline.txt"
