#!/bin/bash
# fresh-tuned bench runs under different environment settings, alternating on one box:
#   tools/env_ab.sh 3 "MEBT_GEMM_PREFETCH=0" "MEBT_GEMM_PREFETCH=1" "MEBT_GEMM_PREFETCH=2"
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
R=$1; shift
run() { env $1 MEBT_GEMM_TUNE_CACHE=/tmp/tc_$RANDOM.txt python3 bench.py --steps 30 --warmup 5 --secondary none --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for i in $(seq 1 $R); do line=""; for s in "$@"; do line="$line  [$s] $(run "$s")"; done; echo "$line"; done
