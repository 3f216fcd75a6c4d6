#!/bin/bash
# A/B of two environment settings of bench.py on ONE box, interleaved rounds (rule: never compare builds across boxes).
#   tools/ab_bench.sh "<env A>" "<env B>" [rounds] [steps]      e.g.  tools/ab_bench.sh "MEBT_BIAS_IN_WGRAD=0" "MEBT_BIAS_IN_WGRAD=1" 3 30
A="$1"; B="$2"; R=${3:-3}; S=${4:-30}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export MEBT_GEMM_TUNE_CACHE=${MEBT_GEMM_TUNE_CACHE:-/tmp/ab_tune_cache.txt}
run() { env $1 python3 bench.py --steps $S --warmup 8 --secondary none --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['gemm_ms_per_step'])"; }
echo "populate: $(run "$A")  $(run "$B")"
for i in $(seq 1 $R); do echo "A [$A]: $(run "$A")    B [$B]: $(run "$B")"; done
