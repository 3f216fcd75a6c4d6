#!/bin/bash
# A/B of two BUILDS of the library on one box (MEBT_HIP_LIB selects another libmebt_hip.so with the same ABI), alternating rounds:
#   tools/lib_ab.sh <other.so> [rounds] [steps]
OTHER=$(readlink -f "$1"); R=${2:-3}; S=${3:-30}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
run() { env $1 python3 bench.py --steps $S --warmup 8 --secondary none --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['gemm_ms_per_step'])"; }
for i in $(seq 1 $R); do echo "other [$1]: $(run "MEBT_HIP_LIB=$OTHER")    in-tree: $(run "X=1")"; done
