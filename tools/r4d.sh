#!/bin/bash
OUT=gpurun_out/r4d; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > $OUT/tests_ops.log 2>&1; echo "ops rc $?" >> $OUT/tests_ops.log; tail -4 $OUT/tests_ops.log
python tools/sample_kernel_bench.py > $OUT/sample_fast.txt 2>&1; MEBT_SAMPLE_FAST=0 python tools/sample_kernel_bench.py > $OUT/sample_old.txt 2>&1
paste -d'|' $OUT/sample_fast.txt $OUT/sample_old.txt
python tools/attn_bench.py > $OUT/attn_auto.txt 2>&1; grep fwd $OUT/attn_auto.txt
timeout 900 python -m pytest tests/test_gpu_product.py tests/test_gpu_model.py tests/test_gpu_dropout.py -m gpu -x -q > $OUT/tests_prod.log 2>&1; echo "prod rc $?" >> $OUT/tests_prod.log; tail -4 $OUT/tests_prod.log
export MEBT_GEMM_TUNE_CACHE=$PWD/$OUT/tune.txt
python bench.py --secondary none --no-cpu-baseline > $OUT/bench_head.json 2> $OUT/bench_head.err; head -c 1200 $OUT/bench_head.json; echo
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/st -o b -- python3 $OLDPWD/bench.py --steps 10 --warmup 5 --secondary none --no-cpu-baseline > /dev/null 2>&1)
cp $OUT/st/*kernel_stats.csv $OUT/c2_kernel_stats.csv; rm -rf $OUT/st; python tools/kernel_table.py $OUT/c2_kernel_stats.csv 16 | tee $OUT/c2_table.txt | tail -12
python bench.py --secondary c4 --c4-legs revise,bootstrap --no-cpu-baseline > $OUT/c4.json 2> $OUT/c4.err; head -c 1500 $OUT/c4.json; echo
python bench.py --secondary c5 --no-cpu-baseline > $OUT/c5.json 2> $OUT/c5.err; head -c 600 $OUT/c5.json; echo
