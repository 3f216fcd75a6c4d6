#!/bin/bash
# round-4 GPU call c: sampler / attention tests on the new kernels, attention forward A/B, merged backward launches (parity + A/B)
OUT=gpurun_out/r4c; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > $OUT/tests_ops.log 2>&1; echo "ops rc $?" >> $OUT/tests_ops.log; tail -4 $OUT/tests_ops.log
for W in 8 4; do MEBT_ATTN_FWD_WAVES=$W python tools/attn_bench.py > $OUT/attn_w$W.txt 2>&1; done
paste -d'|' $OUT/attn_w8.txt $OUT/attn_w4.txt | grep fwd | cut -c1-200
MEBT_BWD_MERGE=1 timeout 1500 python -m pytest tests/test_gpu_benchsize.py tests/test_gpu_model.py tests/test_gpu_dropout.py -m gpu -x -q -k "c2_bf16_train or ragged or model or dropout" > $OUT/tests_merge.log 2>&1; echo "merge rc $?" >> $OUT/tests_merge.log; tail -6 $OUT/tests_merge.log
export MEBT_GEMM_TUNE_CACHE=$PWD/$OUT/tune_ab.txt
export MEBT_GEMM_TUNE_LOG=1
bash tools/ab_bench.sh "MEBT_BWD_MERGE=0" "MEBT_BWD_MERGE=1" 3 30 2> $OUT/ab.err | tee $OUT/ab_merge.txt
grep "multi" $OUT/ab.err | head -20
