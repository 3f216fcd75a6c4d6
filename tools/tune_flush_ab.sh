#!/bin/bash
# fresh-tuned bench runs with the tuner's cache flush at different sizes (384 MB = everything cold incl. the Infinity Cache; 32-64 MB = the
# L2s only), alternating on one box:  tools/tune_flush_ab.sh "64 96 128 192 256 384" 3
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
SIZES=${1:-"64 128 384"}; R=${2:-3}
run() { env $1 MEBT_GEMM_TUNE_CACHE=/tmp/tc_$RANDOM.txt python3 bench.py --steps 30 --warmup 5 --secondary none --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for i in $(seq 1 $R); do line=""; for s in $SIZES; do line="$line  ${s} MB: $(run MEBT_GEMM_TUNE_FLUSH_MB=$s)"; done; echo "$line"; done
