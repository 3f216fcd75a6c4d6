#!/bin/bash
# round-4 GPU call b: new kernels' tests, attention forward 4- vs 8-wave, config-4 legs one by one with the tuner's log
OUT=gpurun_out/r4b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_product.py tests/test_gpu_model.py -m gpu -x -q -k "sampler or next_mask or attention or sample or script_drivers or edit_mode or draft" > $OUT/tests1.log 2>&1; echo "tests1 rc $?" >> $OUT/tests1.log; tail -5 $OUT/tests1.log
for W in 8 4; do MEBT_ATTN_FWD_WAVES=$W python tools/attn_bench.py > $OUT/attn_w$W.txt 2>&1; done
python tools/attn_bench.py > $OUT/attn_auto.txt 2>&1
paste -d'|' $OUT/attn_w8.txt $OUT/attn_w4.txt | grep fwd | cut -c1-200
export MEBT_GEMM_TUNE_CACHE=$PWD/$OUT/tune_c4.txt
for LEG in revise bootstrap train; do
  MEBT_GEMM_TUNE_LOG=1 python bench.py --secondary c4 --c4-legs $LEG --no-cpu-baseline > $OUT/c4_$LEG.json 2> $OUT/c4_$LEG.err
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/st_$LEG -o b -- python3 $OLDPWD/bench.py --secondary c4 --c4-legs $LEG --no-cpu-baseline > $OLDPWD/$OUT/c4_${LEG}_prof.json 2> /dev/null)
  cp $OUT/st_$LEG/*kernel_stats.csv $OUT/c4_${LEG}_kernel_stats.csv; rm -rf $OUT/st_$LEG
  python tools/kernel_table.py $OUT/c4_${LEG}_kernel_stats.csv 22 > $OUT/c4_${LEG}_table.txt
  head -c 700 $OUT/c4_$LEG.json; echo; head -24 $OUT/c4_${LEG}_table.txt
done
grep "autotune" $OUT/c4_*.err | cut -c1-220 > $OUT/c4_tune_log.txt
python bench.py --secondary c5 --no-cpu-baseline > $OUT/c5.json 2> $OUT/c5.err; head -c 1500 $OUT/c5.json
