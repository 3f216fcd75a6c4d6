#!/bin/bash
# in-step tuner refinement, then an A/B of the two tables in separate processes (alternating), tuner off so that nothing changes them
OUT=gpurun_out/r4m; mkdir -p $OUT
MEBT_GEMM_TUNE_SHIPPED=0 python tools/step_tune.py $OUT/refined.txt $OUT/step_tune.log 2> $OUT/err.txt; cat $OUT/step_tune.log
run() { cp $1 /tmp/tc_ab.txt; MEBT_GEMM_TUNE_SHIPPED=0 MEBT_GEMM_TUNE_CACHE=/tmp/tc_ab.txt python bench.py --steps 40 --warmup 8 --secondary none --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['gemm_ms_per_step'])"; }
for i in 1 2 3; do echo "before: $(run $OUT/refined.txt.before)   refined: $(run $OUT/refined.txt)"; done | tee $OUT/ab.txt
