#!/bin/bash
# Kernel trace of the train step under a spatial split of the backward:  tools/profile_split.sh <tag> <MEBT_CU_SPLIT value>
TAG=${1:-split}; N=${2:-16}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export MEBT_GEMM_TUNE_CACHE=$OUT/tune_cache_$N.txt
export MEBT_CU_SPLIT=$N
ARGS="$ROOT/bench.py --steps 10 --warmup 5 --secondary none --no-cpu-baseline"
cd "$ROOT" && python3 $ARGS > "$OUT/populate_$N.json" 2> "$OUT/populate_$N.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$N" -o bench -- python3 $ARGS > "$OUT/stats_$N.json" 2> "$OUT/stats_$N.err"
cd "$ROOT"
cp "$OUT/stats_$N/"*kernel_stats.csv "$OUT/kernel_stats_$N.csv" 2>/dev/null
python3 tools/kernel_table.py "$OUT/kernel_stats_$N.csv" 40 > "$OUT/kernel_table_$N.txt"
python3 tools/trace_timeline.py "$OUT/stats_$N/"*kernel_trace.csv > "$OUT/timeline_$N.txt" 2>&1
rm -rf "$OUT/stats_$N"
head -c 300 "$OUT/stats_$N.json"; echo; cat "$OUT/kernel_table_$N.txt"; head -40 "$OUT/timeline_$N.txt"
