#!/bin/bash
# Per-round profiling recipe (run on the GPU box through gpurun):  tools/profile_round.sh <tag>
#   1. populate the GEMM tune cache (so the profiled runs launch no tuning candidates),
#   2. rocprofv3 --kernel-trace --stats of `bench.py` (20 steps in all: 1 priming (tuner) + 5 warm-up + 10 timed + 2 host-enqueue-timed + 2 event-profiled; the round-3 CSV predates the priming step: 19),
#   3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy cycles, SQ wait counters) as MI355X_MICROARCH.md prescribes.
# Summaries land in gpurun_out/<tag>/; copy what is to be judged into profiles/.
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export MEBT_GEMM_TUNE_CACHE=$OUT/tune_cache.txt
ARGS="$ROOT/bench.py --steps 10 --warmup 5 --windows 1 --secondary none --no-cpu-baseline"      # one window: no clock probe / stream probing kernels in the trace
cd "$ROOT" && python3 $ARGS > "$OUT/populate.json" 2> "$OUT/populate.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 $ARGS > "$OUT/stats.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o b -- python3 $ARGS > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o b -- python3 $ARGS > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -o b -- python3 $ARGS > "$OUT/pmc_mfma.json" 2> "$OUT/pmc_mfma.err"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/pmc_sq" -o b -- python3 $ARGS > "$OUT/pmc_sq.json" 2> "$OUT/pmc_sq.err"
cd "$ROOT"
python3 tools/pmc_sq.py "$OUT/pmc_sq" "$OUT/pmc_sq_breakdown.json" > "$OUT/pmc_sq_breakdown.txt" 2>&1
python3 tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.txt" 2>&1
python3 tools/pmc_mfma.py "$OUT/pmc_mfma" "$OUT/pmc_mfma_summary.json" > "$OUT/pmc_mfma.txt" 2>&1
# keep the merge-back small: the raw counter CSVs are tens of MB
cp "$OUT/stats/"*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_mfma" "$OUT/pmc_sq"
tail -3 "$OUT/pmc_traffic.txt" "$OUT/pmc_mfma.txt"; head -c 600 "$OUT/stats.json"
