#!/bin/bash
# Like env_ab.sh, but every setting first fills a tune cache of its own (one untimed run: in-situ tuning), then the settings are timed in
# R alternating rounds from those caches (the process that tunes is not the process that is timed):
#   tools/env_ab_cached.sh 3 "MEBT_CU_SPLIT=0" "MEBT_CU_SPLIT=16" ...
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
R=$1; shift
STEPS=${STEPS:-30}
run() { env $(echo $1 | tr "," " ") MEBT_GEMM_TUNE_CACHE=$2 python3 bench.py --steps $STEPS --warmup 5 --secondary none --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['gemm_ms_per_step'])"; }
i=0; for s in "$@"; do i=$((i+1)); rm -f /tmp/tcc_$i.txt; echo "populate [$s] $(run "$s" /tmp/tcc_$i.txt)"; done
for r in $(seq 1 $R); do line=""; i=0; for s in "$@"; do i=$((i+1)); line="$line  [$s] $(run "$s" /tmp/tcc_$i.txt)"; done; echo "$line"; done
